"""Prints the kernels that follow the n-th-from-last launch of an anchor kernel in a rocprofv3 --kernel-trace database:
start offset, duration, queue -- to see what overlaps what."""
import sqlite3
import sys


def main(db, anchor="k_luts_tables", back=3, count=10):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    ev = sorted(c.execute("select start, end, name, %s from kernels" % qcol))
    idx = [i for i, e in enumerate(ev) if anchor in e[2]]
    if len(idx) < back:
        raise SystemExit("anchor %r found %d times" % (anchor, len(idx)))
    i0 = idx[-back]
    t0 = ev[i0][0]
    for s, e, name, q in ev[max(i0 - 2, 0):i0 + count]:
        name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
        print("  +%8.1f us  %7.1f us  q=%-6s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, name))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2] if len(a) > 2 else "k_luts_tables", int(a[3]) if len(a) > 3 else 3, int(a[4]) if len(a) > 4 else 10)

"""Which zero-fills does a step pay for?  rocprofv3 --kernel-trace database -> the fill dispatches of ONE step with their duration and the
kernels launched right before / after each (the consumer names the buffer).    python scripts/fill_trace.py <db> [min_us]"""
import sqlite3
import sys


def short(n):
    for junk in ("(anonymous namespace)::", "void "):
        n = n.replace(junk, "")
    return n.split("(")[0][:60]


def main(db, min_us=20.0):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end, grid_x, stream_id from kernels order by start"))
    fills = [i for i, r in enumerate(rows) if "fillBuffer" in r[0]]
    tot = sum(rows[i][2] - rows[i][1] for i in fills) / 1e3
    print("%d fills, %.1f us in total" % (len(fills), tot))
    seen = {}
    for i in fills:
        r = rows[i]
        us = (r[2] - r[1]) / 1e3
        if us < min_us:
            continue
        prev = short(rows[i - 1][0]) if i else "-"
        nxt = [short(rows[j][0]) for j in range(i + 1, min(i + 4, len(rows)))]
        key = (r[3], prev, tuple(nxt))
        a = seen.setdefault(key, [0, 0.0])
        a[0] += 1; a[1] += us
    for (g, prev, nxt), (n, us) in sorted(seen.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%5d x %8.1f us  grid %-10d after %-40s before %s" % (n, us / n, g, prev, " | ".join(nxt)))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 20.0)

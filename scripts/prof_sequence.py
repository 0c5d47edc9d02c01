"""rocprofv3 rocpd database -> time-ordered, run-length compressed kernel sequence of the last full step
(a steady-state window between two consecutive launches of the marker kernel, default k_fused)."""
import sqlite3
import sys


def main(db, marker="k_fused"):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end, grid_x, grid_y, grid_z from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    lo, hi = 0, len(rows)
    if len(idx) >= 2:
        # marker-to-marker windows; a full step is much longer (in kernels) than a hot-path iteration, and the first full step
        # also carries lazy initialisations: take the LAST window among those whose kernel count is within 2 % of the median
        # count of the long windows
        spans = [(idx[i], idx[i + 1]) for i in range(len(idx) - 1)]
        longest = max(b - a for a, b in spans)
        full = sorted(b - a for a, b in spans if (b - a) * 2 > longest)
        med = full[len(full) // 2]
        steady = [(a, b) for a, b in spans if abs((b - a) - med) <= 0.02 * med]
        lo, hi = steady[-1] if steady else max(spans, key=lambda ab: ab[1] - ab[0])
    out, prev, cnt, dur = [], None, 0, 0.0
    for r in rows[lo:hi]:
        name = r[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
        key = (name, r[3], r[4], r[5])
        if key == prev:
            cnt += 1; dur += (r[2] - r[1]) / 1e3
        else:
            if prev is not None:
                out.append("%4d x %-90s %10.1f us  grid %dx%dx%d" % (cnt, prev[0], dur, prev[1], prev[2], prev[3]))
            prev, cnt, dur = key, 1, (r[2] - r[1]) / 1e3
    if prev is not None:
        out.append("%4d x %-90s %10.1f us  grid %dx%dx%d" % (cnt, prev[0], dur, prev[1], prev[2], prev[3]))
    print("\n".join(out))
    print("window: %.2f ms wall, %.2f ms busy" % ((rows[hi - 1][2] - rows[lo][1]) / 1e6, sum(r[2] - r[1] for r in rows[lo:hi]) / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:])

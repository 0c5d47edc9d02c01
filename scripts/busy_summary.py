"""GPU busy fraction of the steady-state part of a rocprofv3 --kernel-trace run: union of kernel intervals / span over the
last `frac` of the trace (default 0.5), plus launches in that window.  Says whether a small-batch step is host- or GPU-bound."""
import sqlite3
import sys


def main(db, frac=0.5):
    c = sqlite3.connect(db)
    ev = sorted(c.execute("select start, end from kernels"))
    t0, t1 = ev[0][0], ev[-1][1]
    cut = t1 - (t1 - t0) * frac
    ev = [e for e in ev if e[0] >= cut]
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    gaps = []
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = ev[-1][1] - ev[0][0]
    gaps.sort()
    print("window %.1f ms: %d launches, busy %.1f ms (%.1f%%), sum of kernel durations %.1f ms" % (
        span / 1e6, len(ev), busy / 1e6, 100.0 * busy / span, sum(e - s for s, e in ev) / 1e6))
    if gaps:
        n = len(gaps)
        print("gaps: n %d, median %.1f us, p90 %.1f us, p99 %.1f us, total %.1f ms; gaps > 20 us: %d (%.1f ms)" % (
            n, gaps[n // 2] / 1e3, gaps[int(n * 0.9)] / 1e3, gaps[int(n * 0.99)] / 1e3, sum(gaps) / 1e6,
            sum(1 for g in gaps if g > 20000), sum(g for g in gaps if g > 20000) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)

"""GPU busy fraction of a window of a rocprofv3 --kernel-trace run: union of kernel intervals / span, launches, gap statistics.
The window runs from the (skip+1)-th to the (skip+count+1)-th launch of a marker kernel that occurs once per step (default:
the stem convolution), i.e. exactly `count` steps.  Says whether a small-batch step is host- or GPU-bound."""
import sqlite3
import sys


def main(db, marker="k_stem7x7<", skip=8, count=20):
    c = sqlite3.connect(db)
    ev = sorted(c.execute("select start, end, name from kernels"))
    marks = [e[0] for e in ev if e[2].replace("(anonymous namespace)::", "").replace("void ", "").startswith(marker)]
    if len(marks) < skip + count + 1:
        raise SystemExit("marker %r occurs %d times, need %d" % (marker, len(marks), skip + count + 1))
    lo, hi = marks[skip], marks[skip + count]
    ev = [e for e in ev if lo <= e[0] < hi]
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    gaps = []
    for s, e, _ in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = hi - lo
    gaps.sort()
    n = max(len(gaps), 1)
    print("%d steps: %.2f ms per step, %d launches per step, busy %.2f ms per step (%.1f%%)" % (
        count, span / 1e6 / count, len(ev) // count, busy / 1e6 / count, 100.0 * busy / span))
    print("gaps per step: %d, total %.2f ms; median %.1f us, p90 %.1f us, p99 %.1f us; > 20 us: %.1f per step = %.2f ms" % (
        len(gaps) // count, sum(gaps) / 1e6 / count, gaps[n // 2] / 1e3, gaps[int(n * 0.9)] / 1e3, gaps[int(n * 0.99)] / 1e3,
        sum(1 for g in gaps if g > 20000) / count, sum(g for g in gaps if g > 20000) / 1e6 / count))
    # the longest gaps and what ran before / after them
    big = []
    cur_e, prev = ev[0][1], ev[0][2]
    for s, e, name in ev[1:]:
        if s > cur_e:
            big.append((s - cur_e, prev, name))
        if e > cur_e:
            cur_e, prev = e, name
    big.sort(reverse=True)
    for g, a, b in big[:12]:
        print("  gap %7.1f us after %-40s before %s" % (g / 1e3, a.replace("(anonymous namespace)::", "")[:40], b.replace("(anonymous namespace)::", "")[:60]))


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2] if len(a) > 2 else "k_stem7x7<", int(a[3]) if len(a) > 3 else 8, int(a[4]) if len(a) > 4 else 20)

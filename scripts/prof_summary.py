"""Turns a rocprofv3 rocpd database (--kernel-trace) into a per-kernel text summary (for profiles/)."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(grid_y), max(grid_z), max(workgroup_x) "
        "from kernels group by name order by 6 desc"))
    total = sum(r[5] for r in rows) or 1
    lines = ["%-72s %6s %10s %10s %10s %10s %6s  vgpr sgpr   lds  grid" % ("kernel", "calls", "avg_us", "min_us", "max_us", "total_ms", "pct")]
    for r in rows:
        name = r[0]
        for junk in ("(anonymous namespace)::", "void "):
            name = name.replace(junk, "")
        name = name.split("(")[0][:72]
        lines.append("%-72s %6d %10.1f %10.1f %10.1f %10.2f %5.1f%%  %4d %4d %5d  %dx%dx%d/%d" % (
            name, r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e6, 100.0 * r[5] / total, r[6], r[7], r[8],
            r[9], r[10], r[11], r[12]))
    text = "\n".join(lines)
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)

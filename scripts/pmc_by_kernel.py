"""rocprofv3 --pmc rocpd database(s) under a directory -> one line per (kernel, counter): average value and dispatch count.
    python pmc_by_kernel.py <dir> [substring filter]"""
import glob
import sqlite3
import sys


def main(root, needle=""):
    for db in glob.glob(root + "/**/*.db", recursive=True):
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" not in tabs:
            print("tables:", tabs[:40])
            continue
        cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        kn = "kernel_name" if "kernel_name" in cols else "name"
        q = ("select %s, counter_name, avg(value), count(*) from counters_collection where %s like ? "
             "group by %s, counter_name order by 1, 2" % (kn, kn, kn))
        for name, ctr, avg, n in c.execute(q, ("%" + needle + "%",)):
            for junk in ("(anonymous namespace)::", "void "):
                name = name.replace(junk, "")
            print("%-60s %-24s avg %.1f over %d dispatches" % (name.split("(")[0][:60], ctr, avg, n))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")

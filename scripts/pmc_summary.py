"""rocprofv3 --pmc rocpd database(s) under a directory -> average counter values per kernel (substring filter)."""
import glob
import sqlite3
import sys


def main(root, needle):
    dbs = glob.glob(root + "/**/*.db", recursive=True)
    if not dbs:
        print("no .db under", root)
        return
    for db in dbs:
        c = sqlite3.connect(db)
        tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
        if "counters_collection" in tabs:       # rocpd view: one row per (dispatch, counter)
            cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
            kn = "kernel_name" if "kernel_name" in cols else "name"
            q = "select counter_name, avg(value), count(*) from counters_collection where %s like ? group by counter_name" % kn
            for r in c.execute(q, ("%" + needle + "%",)):
                print("%-28s avg %.1f over %d dispatches" % r)
        else:
            print("tables:", tabs[:40])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_fused<16>")

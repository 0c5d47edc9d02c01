import json, sqlite3, sys
meta = json.load(open(sys.argv[2]))
c = sqlite3.connect(sys.argv[1])
rows = [r for r in c.execute("select name, start, end from kernels order by start") if "k_hist_fused" in r[0]]
R = meta["R"]
assert len(rows) == R * len(meta["order"]), (len(rows), R, len(meta["order"]))
for i, label in enumerate(meta["order"]):
    d = [(r[2] - r[1]) / 1e3 for r in rows[i * R:(i + 1) * R]]
    print("%-32s k_hist_fused %7.1f us (min %6.1f)" % (label, sum(d[1:]) / (R - 1), min(d)))

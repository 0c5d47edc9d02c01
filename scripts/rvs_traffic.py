"""profiles/r04_rvs1024_traffic.json: HBM bytes per dispatch of the tile kernels of the 1024 x 1024 RVS leg from the two PMC passes
(FETCH_SIZE doubled on gfx950, WRITE_SIZE as is: MI355X_MICROARCH.md, HBM section).
    python rvs_traffic.py rvs1024_leg.json pmc_rvs1024_FETCH_SIZE.txt pmc_rvs1024_WRITE_SIZE.txt"""
import json
import re
import sys


def table(path):
    t = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+avg\s+([\d.]+)\s+over\s+(\d+)", line)
        if m:
            t[m.group(1).strip()] = (float(m.group(3)), int(m.group(4)))
    return t


leg = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["rvs_1024"]
f, w = table(sys.argv[2]), table(sys.argv[3])
tile = [k for k in sorted(set(f) | set(w)) if k.startswith(("k_fused", "k_gen_"))]
out = {"command": "rocprofv3 --pmc FETCH_SIZE -- python bench.py --only_legs rvs1024  (and a separate pass with --pmc WRITE_SIZE)",
       "correction": "FETCH_SIZE (KB) x 2 on gfx950, WRITE_SIZE (KB) as is", "workload": leg["workload"], "units": leg["units"], "kernels": {}}
total = 0
batches = max(f.get("k_fused3", w.get("k_fused3", (0.0, 1)))[1], 1)          # k_fused3 runs once per batch; the two passes once per chunk of units
for k in tile:
    fk, wk = f.get(k, (0.0, 0))[0], w.get(k, (0.0, 0))[0]
    n = f.get(k, w.get(k))[1]
    b = int(2 * fk * 1024 + wk * 1024)
    per_batch = b * n // batches
    total += per_batch
    out["kernels"][k] = {"fetch_bytes": int(2 * fk * 1024), "write_bytes": int(wk * 1024), "hbm_bytes_per_dispatch": b, "dispatches": n,
                         "hbm_bytes_per_batch": per_batch}
out["batches"] = batches
out["tile_kernels_hbm_bytes_per_batch"] = total
out["hbm_bytes_per_unit"] = total // max(leg["units"], 1)
out["algorithmic_bytes_per_batch"] = leg["roofline"]["bytes_per_launch"]
out["note"] = ("traffic / algorithmic bytes = %.2f: the packed intermediate of the down-scaling units (written by k_gen_hpass, read by k_gen_vpass) "
               "is extra traffic the one-pass tile did not have -- as far as it leaves the L2s: the chunks' intermediate is sized to stay in the Infinity Cache, which these counters (L2 <-> fabric) cannot see" %
               (total / leg["roofline"]["bytes_per_launch"]))
print(json.dumps(out, indent=1))

"""profiles/r06_rvs1024_traffic.json (round 6: + units per tile kernel and each kernel group's fraction of the HBM peak, so that the fractions DESIGN.md
quotes reproduce from this file; argv[4] = the rocprofv3 kernel table of the same leg): HBM bytes per dispatch of the tile kernels of the 1024 x 1024 RVS leg from the two PMC passes
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
out["file"] = "r06_rvs1024_traffic.json"
out["batches"] = batches
# per kernel group: units per batch (bench.py leg: units_per_batch_by_tile_kernel), algorithmic bytes (6 MiB in + 16 MiB out at 1024 x 1024,
# K = 1: 3 source bytes + 1 mask byte per source pixel, 4 float32 planes per output pixel), duration per batch from the kernel table
tu = leg.get("units_per_batch_by_tile_kernel")
if tu and len(sys.argv) > 4:
    dur = {}
    for line in open(sys.argv[4]):
        p = line.split()
        if p and p[0].startswith(("k_fused", "k_gen_")):
            nums = [x for x in p if x.replace(".", "", 1).isdigit()]
            dur[p[0]] = (int(nums[0]), float(nums[1]))                      # calls, average us
    per_unit = leg["roofline"]["bytes_per_launch"] / leg["units"]
    groups = {"k_fused3": ["k_fused3"], "k_fused3w": ["k_fused3w"], "two_pass": ["k_gen_hpass<false>", "k_gen_hpass<true>", "k_gen_vpass"]}
    out["groups"] = {}
    for gname, ks in groups.items():
        us = sum(dur[k][0] * dur[k][1] for k in ks if k in dur) / batches
        hbm = sum(out["kernels"][k]["hbm_bytes_per_batch"] for k in ks if k in out["kernels"])
        units = tu[gname]
        alg = units * per_unit
        out["groups"][gname] = {"units_per_batch": units, "kernel_us_per_batch": us, "algorithmic_bytes_per_batch": int(alg),
                                "frac_of_8TBps": alg / (us * 1e-6) / 8e12 if us else None, "hbm_bytes_per_batch": hbm,
                                "traffic_over_algorithmic": hbm / alg if alg else None}
out["tile_kernels_hbm_bytes_per_batch"] = total
out["hbm_bytes_per_unit"] = total // max(leg["units"], 1)
out["algorithmic_bytes_per_batch"] = leg["roofline"]["bytes_per_launch"]
out["note"] = ("traffic / algorithmic bytes = %.2f: the packed intermediate of the down-scaling units (written by k_gen_hpass, read by k_gen_vpass) "
               "is extra traffic the one-pass tile did not have -- as far as it leaves the L2s: the chunks' intermediate is sized to stay in the Infinity Cache, which these counters (L2 <-> fabric) cannot see" %
               (total / leg["roofline"]["bytes_per_launch"]))
print(json.dumps(out, indent=1))

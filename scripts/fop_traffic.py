"""profiles/r04_fop_traffic.json: HBM bytes per dispatch of every k_fop_* kernel from the two PMC passes (FETCH_SIZE doubled on
gfx950, WRITE_SIZE as is: MI355X_MICROARCH.md, HBM section) next to the bytes the float_ops leg prices the op at.
    python fop_traffic.py fop_leg.json pmc_fop_FETCH_SIZE.txt pmc_fop_WRITE_SIZE.txt"""
import json
import re
import sys


def table(path):
    t = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+avg\s+([\d.]+)\s+over\s+(\d+)", line)
        if m and "k_fop" in m.group(1):
            t[m.group(1).strip()] = (float(m.group(3)), int(m.group(4)))
    return t


leg = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["float_ops"]
f, w = table(sys.argv[2]), table(sys.argv[3])
out = {"command": "rocprofv3 --pmc FETCH_SIZE -- python bench.py --only_legs fop  (and a separate pass with --pmc WRITE_SIZE)",
       "correction": "FETCH_SIZE (KB) x 2 on gfx950, WRITE_SIZE (KB) as is", "workload": leg["workload"], "kernels": {}}
for k in sorted(set(f) | set(w)):
    fk, wk = f.get(k, (0.0, 0))[0], w.get(k, (0.0, 0))[0]
    out["kernels"][k] = {"fetch_bytes": int(2 * fk * 1024), "write_bytes": int(wk * 1024), "hbm_bytes_per_dispatch": int(2 * fk * 1024 + wk * 1024),
                         "dispatches": f.get(k, w.get(k))[1]}
B, H = 144, 512
out["one_image_pass_bytes"] = {"read_or_write_of_the_batch": 12 * H * H * B, "24HW_B": 24 * H * H * B, "36HW_B": 36 * H * H * B}
print(json.dumps(out, indent=1))

import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["rvs_1024"]
r = d["roofline"]
print(d["units_by_tile_kernel"], "kernel_ms %.3f frac %.3f stage_ms %.3f stage_frac %.3f" % (r["kernel_ms"], r["frac"], r["stage"]["ms"], r["stage"]["frac"]))

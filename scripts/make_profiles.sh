# Regenerates the round-2 profile artefacts on a GPU box (run through gpurun; results land in gpurun_out/r2/profiles/, copy the
# ones to keep into profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/profiles
mkdir -p $O
# 1. per-kernel durations of the bench command
rocprofv3 --kernel-trace --stats -d /tmp/prof_b -- python $R/bench.py --legs none --steps 10 --warmup 3 > $O/r02_bench_under_rocprof.json 2>/dev/null
DB=$(find /tmp/prof_b -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB $O/r02_bench_rocprofv3_kernel_stats.txt > /dev/null
python $R/scripts/busy_summary.py $DB "k_stem7x7<" 3 8 > $O/r02_bench_busy.txt
# 2. HBM traffic of the dominant kernel (separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pmc_$c -- python $R/bench.py --legs none --steps 5 --warmup 2 > /tmp/bench_$c.json 2>/dev/null
  python $R/scripts/pmc_summary.py /tmp/pmc_$c "k_fused3" > $O/r02_pmc_$c.txt
done
# 3. the bench line itself (all legs), not under a profiler
python $R/bench.py > $O/r02_bench_n1.json 2> $O/r02_bench_n1.err
# 4. one rank of an 8-rank job (18 + 3 rows), kernel statistics and busy fraction
rocprofv3 --kernel-trace --stats -d /tmp/prof_s8 -- python $R/bench.py --legs none --shard_of 8 --steps 30 --warmup 5 > $O/r02_shard8_under_rocprof.json 2>/dev/null
DB8=$(find /tmp/prof_s8 -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB8 $O/r02_shard8_rocprofv3_kernel_stats.txt > /dev/null
python $R/scripts/busy_summary.py $DB8 "k_stem7x7<" 8 20 > $O/r02_shard8_busy.txt
python $R/bench.py --legs none --shard_of 8 --steps 30 --warmup 5 > $O/r02_shard8.json 2>/dev/null
# 5. the derived JSON files bench.py quotes
python - <<PY
import json, re
O = "$O"
def avg(name):
    for l in open(O + "/r02_bench_rocprofv3_kernel_stats.txt"):
        if l.startswith(name + " "):
            p = l.split()
            return int(p[1]), float(p[2]) / 1e3
    return 0, 0.0
stats = {"file": "r02_bench_rocprofv3_kernel_stats.txt",
         "command": "rocprofv3 --kernel-trace --stats -- python bench.py --legs none --steps 10 --warmup 3"}
nf = avg("k_fused3")[0]
for k in ("k_fused3", "k_luts_tables", "k_hist_fused", "k_lut"):
    n, a = avg(k)
    stats[k + "_avg_ms"] = round(a, 4)
    stats[k + "_calls_per_launch"] = round(n / max(nf, 1), 2)
json.dump(stats, open(O + "/r02_bench_kernel_stats.json", "w"))
def pmc(c):
    t = open(O + "/r02_pmc_%s.txt" % c).read()
    return float(re.search(r"avg\s+([\d.]+)", t).group(1))
b = json.load(open("/tmp/bench_FETCH_SIZE.json"))["roofline"]
f, w, units = pmc("FETCH_SIZE"), pmc("WRITE_SIZE"), b["units_per_launch"]
hbm = int(2 * f * 1024 + w * 1024)
json.dump({"kernel": "k_fused3", "size": 512, "units_measured": units,
           "command": "rocprofv3 --pmc FETCH_SIZE -- python bench.py --legs none --steps 5 --warmup 2   (and a second, separate pass with --pmc WRITE_SIZE); scripts/make_profiles.sh",
           "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w,
           "correction": "gfx950 FETCH_SIZE counts 64 B per 128 B request: doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE used as is",
           "hbm_bytes_per_launch": hbm, "hbm_bytes_per_unit": hbm // units, "kernel_bytes": b["bytes_per_launch"],
           "note": "traffic / bytes the kernel addresses = %.3f: the 24 source images of the pool are each read by ~7 units and partly served by the 256 MiB Infinity Cache; no wasted re-reads. bench.py scales hbm_bytes_per_unit by the units of its launch" % (hbm / b["bytes_per_launch"])},
          open(O + "/r02_traffic_k_fused3.json", "w"), indent=2)
PY
grep -n "k_fused3\|k_hist\|k_lut\|k_sinkhorn\|k_embed\|k_seg\|k_ctrl" $O/r02_bench_rocprofv3_kernel_stats.txt | cut -c1-150
cat $O/r02_pmc_FETCH_SIZE.txt $O/r02_pmc_WRITE_SIZE.txt $O/r02_bench_busy.txt $O/r02_shard8_busy.txt
python -c "
import json
for f in ['r02_bench_n1','r02_shard8']:
    d=json.loads(open('$O/%s.json'%f).read().strip().splitlines()[-1]);r=d['roofline'];print(f, d['ms_per_step'], d['value'], r['frac'], r['kernel_ms'], r['stage']['frac'], r['stage']['ms'], d['hot_path']['ms_per_step'])
d=json.loads(open('$O/r02_bench_n1.json').read().strip().splitlines()[-1]); print(json.dumps(d.get('rvs_1024',{}).get('roofline',{}))[:400]); print(d.get('fp32_backbone',{}).get('ms_per_step')); print(json.dumps(d.get('cpu_baseline'))[:600])
"

# HBM traffic of k_fused3 on bench.py's own launch: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (MI355X_MICROARCH.md, HBM section)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pmc_$c -- python $R/bench.py --legs none --steps 5 --warmup 2 > /tmp/bench_$c.json 2>/dev/null
  python $R/scripts/pmc_summary.py /tmp/pmc_$c "k_fused3" | tee $R/gpurun_out/r2/pmc_$c.txt
done
python - <<PY
import json
d=json.load(open("/tmp/bench_FETCH_SIZE.json")); print("units", d["roofline"]["units_per_launch"], "bytes", d["roofline"]["bytes_per_launch"])
PY

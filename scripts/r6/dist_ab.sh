# plain step vs the one-rank run of the distributed path (own reducer, side stream on): bash scripts/r6/dist_ab.sh [pairs]
# writes gpurun_out/r6/dist_ab.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/r6
N=${1:-3}
out=gpurun_out/r6/dist_ab.txt; : > $out
for i in $(seq $N); do
  for v in plain dist; do
    extra=""; [ $v = dist ] && extra="--force_dist --dist_backend nccl"
    python bench.py --gpus 1 --steps 10 --warmup 3 --legs none $extra 2> gpurun_out/r6/dist_ab_$v.err | grep '^{' | tail -1 > /tmp/line.json
    python - "$v" >> $out <<'PY'
import json,sys
d=json.load(open('/tmp/line.json'))
print(sys.argv[1], "ms_per_step %.2f" % d["ms_per_step"], json.dumps(d["config"]["distributed"].get("gradient_buckets")), d["config"]["distributed"].get("weight_gradient_side_stream"))
PY
  done
done
cat $out

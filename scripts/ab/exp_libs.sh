# A/B of library variants in exp_libs/*.so on the GPU box: rvs1024 leg + headline kernel, one line each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
for v in "$@"; do
  cp exp_libs/$v.so aadg_amd/lib/libaadg_hip.so
  python bench.py --only_legs rvs1024 > gpurun_out/exp/rvs_$v.json 2> gpurun_out/exp/rvs_$v.err
  python bench.py --legs none --steps 10 --warmup 3 > gpurun_out/exp/bench_$v.json 2> gpurun_out/exp/bench_$v.err
  python - <<PY
import json
r = json.load(open("gpurun_out/exp/rvs_$v.json"))["rvs_1024"]["roofline"]
b = json.loads([l for l in open("gpurun_out/exp/bench_$v.json") if l.startswith("{")][-1])
print("$v", "rvs kernel_ms %.4f frac %.3f stage_ms %.4f | k_fused3 ms %.4f frac %.3f stage ms %.4f | step ms %.2f" % (
    r["kernel_ms"], r["frac"], r["stage"]["ms"], b["roofline"]["kernel_ms"], b["roofline"]["frac"], b["roofline"]["stage"]["ms"], b["ms_per_step"]))
PY
done

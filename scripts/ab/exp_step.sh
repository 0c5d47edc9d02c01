# the headline step (no extra legs) with library variants exp_libs/<tag>.so, one line each: bash scripts/ab/exp_step.sh <tag>...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
for v in "$@"; do
  cp exp_libs/$v.so aadg_amd/lib/libaadg_hip.so
  python bench.py --legs none --steps 15 --warmup 3 > gpurun_out/exp/step_$v.json 2> gpurun_out/exp/step_$v.err
  python -c "
import json
b=json.loads([l for l in open('gpurun_out/exp/step_$v.json') if l.startswith('{')][-1])
print('$v', 'ms_per_step %.2f' % b['ms_per_step'], b['step_ms'])"
done

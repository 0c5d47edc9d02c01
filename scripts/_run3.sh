mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_controller_fused.py tests/test_gpu_search_loop.py tests/test_gpu_bench_multirank.py -x -q 2>&1 | tail -5
python scripts/time_controller.py
AADG_CTRL_GENERIC=1 python -m pytest tests/test_gpu_controller_fused.py -x -q 2>&1 | tail -2
python bench.py --legs none > gpurun_out/r2/bench_ctrl.json 2>gpurun_out/r2/bench_ctrl.err
python -c "
import json;d=json.loads(open('gpurun_out/r2/bench_ctrl.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['stage']['frac'], d['hot_path'])"
python bench.py --legs none --shard_of 8 --steps 30 --warmup 5 > gpurun_out/r2/shard8b.json 2>/dev/null
python -c "
import json;d=json.loads(open('gpurun_out/r2/shard8b.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'], d['hot_path'])"

mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_controller_fused.py tests/test_gpu_search_loop.py -x -q 2>&1 | tail -5
python scripts/time_controller.py
AADG_LIB_PATH=$PWD/scripts/ubench/libaadg_timed.so python scripts/ubench/ctrl_phase_times.py

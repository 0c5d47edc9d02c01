"""Ad-hoc: cProfile of the host side of bench.py's timed loop (where does the Python time of one step go; waits on the GPU show
up under Tensor.cpu / synchronize)."""
import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--no_cpu_baseline", "--steps", "12", "--warmup", "3"] + sys.argv[1:]
import bench
pr = cProfile.Profile()
orig = bench.time.perf_counter
state = {"on": False}
import torch
real_step = None
def run():
    pr.enable(); bench.main(); pr.disable()
run()
s = io.StringIO()
st = pstats.Stats(pr, stream=s); st.sort_stats("tottime").print_stats(28); st.print_callers("_cuda_getDeviceCount"); st.print_callers("is_available")
print(s.getvalue()[:12000], file=sys.stderr)

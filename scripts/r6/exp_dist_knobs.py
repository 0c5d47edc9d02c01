"""A/B knobs for the one-rank-over-RCCL step: python scripts/r6/exp_dist_knobs.py <fork 0|1> <cdraw 0|1> [bench args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aadg_amd import _lib
from aadg_amd.data import transform as T
_lib.AUG_FORK = sys.argv[1] == "1"
T.USE_C_DRAW = sys.argv[2] == "1"
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[3:]
runpy.run_path(sys.argv[0], run_name="__main__")

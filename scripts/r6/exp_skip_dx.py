"""Upper bound of folding the lazy layers' k_bn_dx into their consumers (timing only, numerics wrong): python scripts/r6/exp_skip_dx.py [0|1] [bench args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aadg_amd import _lib
_lib._EXP["skip_lazy_dx"] = sys.argv[1] == "1"
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")

"""Kernel A/B experiments only (scripts/ab/*.sh, scripts/ubench/*): with this directory on PYTHONPATH and AADG_LIB_PATH set, the
package's binding loads that library variant (scripts/ab/build_variant.py) instead of aadg_amd/lib/libaadg_hip.so.  The product
loader itself reads no environment variable (hygiene): the override lives here, outside the package."""
import os
import sys

_p = os.environ.get("AADG_LIB_PATH")
if _p:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
    from aadg_amd import _lib
    _lib.LIB_PATH = os.path.abspath(_p)
    print("scripts/ab/hook: aadg_amd._lib.LIB_PATH = %s" % _lib.LIB_PATH, file=sys.stderr)
_x = os.environ.get("AADG_AB_EXEC")        # a python statement run at start-up (A/Bs of a host-side switch), e.g.
if _x:                                     # AADG_AB_EXEC="import torch, aadg_amd.models.deeplab as d; d._INPLACE_CONCAT_DTYPES = (torch.bfloat16,)"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
    exec(_x)
    print("scripts/ab/hook: ran AADG_AB_EXEC", file=sys.stderr)


# the stem's BatchNorm backward with the pooling gradient rebuilt on the fly (float32, default) against maxpool backward + BatchNorm backward
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
python bench.py --legs none --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused', r['ms_per_step'])"
python -c "
import sys; sys.argv=['bench.py','--legs','none','--steps','8','--warmup','3']
import torch
from aadg_amd import _lib
orig = _lib._BNReluMaxPool.backward
import runpy
_lib._POOL_BWD_FUSED_F32 = False
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('three passes', r['ms_per_step'])"
done

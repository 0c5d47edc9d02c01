# bn1 / bn2 + ReLU on the consuming convolution's operand load: both (default) / bn2 only / neither, interleaved on ONE box
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
for v in both bn2 none; do
python -c "
import sys; sys.argv=['bench.py','--legs','none','--steps','8','--warmup','3']
from aadg_amd.models import deeplab
v = '$v'
deeplab.Bottleneck.lazy_bn1 = v == 'both'
deeplab.Bottleneck.lazy_bn2 = v != 'none'
import runpy; runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', r['ms_per_step'])"
done
done

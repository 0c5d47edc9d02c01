# BatchNorm on the consumer's operand load: all (default: bn1 / bn2 of the bottlenecks, the decoder's last one, the projection shortcuts') /
# the shortcuts' backward on its own (nopair) / all but the shortcuts / none, interleaved on ONE box
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
for v in all nopair noshort none; do
python -c "
import sys; sys.argv=['bench.py','--legs','none','--steps','8','--warmup','3']
from aadg_amd.models import deeplab
v = '$v'
deeplab.Bottleneck.lazy_bn1 = deeplab.Bottleneck.lazy_bn2 = deeplab.DeepLabV3Plus.lazy_fuse_bn = v != 'none'
deeplab.Bottleneck.lazy_shortcut = v in ('all', 'nopair')
deeplab.Bottleneck.pair_shortcut_bn = v == 'all'
import runpy; runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', r['ms_per_step'])"
done
done

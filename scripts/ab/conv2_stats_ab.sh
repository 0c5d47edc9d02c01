cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
python bench.py --legs none --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('with', r['ms_per_step'])"
python -c "
import sys; sys.argv=['bench.py','--legs','none','--steps','8','--warmup','3']
from aadg_amd import _lib
_lib.conv3x3_x3_stats_supported=lambda *a: False
import runpy; runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('without', r['ms_per_step'])"
done

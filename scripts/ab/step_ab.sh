# the headline step for library variants on ONE box, interleaved: bash scripts/ab/step_ab.sh <reps> [tag ...]   (tree = the built library)
cd $GRAFT_REPO_ROOT
reps=$1; shift
for i in $(seq $reps); do
  for v in ${@:-tree}; do
    if [ "$v" = tree ]; then unset AADG_LIB_PATH; else export AADG_LIB_PATH=$PWD/exp_libs/$v.so PYTHONPATH=$PWD/scripts/ab/hook:$PYTHONPATH; fi
    python bench.py --legs none --steps 8 --warmup 3 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', r['ms_per_step'])"
  done
done

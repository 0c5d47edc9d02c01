# the aug512 and rvs1024 legs for library variants on ONE box: bash scripts/ab/exp_aug.sh [tag ...]  (tree = the built library)
cd $GRAFT_REPO_ROOT
for v in ${@:-tree}; do
  if [ "$v" = tree ]; then unset AADG_LIB_PATH; else export AADG_LIB_PATH=$PWD/exp_libs/$v.so PYTHONPATH=$PWD/scripts/ab/hook:$PYTHONPATH; fi
  for leg in aug512:aug_512 rvs1024:rvs_1024; do
    python bench.py --only_legs ${leg%%:*} 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])['${leg##*:}']['roofline']
print('$v ${leg%%:*}', 'tile kernels %.4f ms frac %.3f | call %.4f ms frac %.3f' % (r['kernel_ms'], r['frac'], r['stage']['ms'], r['stage']['frac']))"
  done
done

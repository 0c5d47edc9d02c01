# depthwise kernels of the step for library variants: bash scripts/ab/dw_prof.sh [tag ...]
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in ${@:-tree}; do
  if [ "$v" = tree ]; then unset AADG_LIB_PATH; else export AADG_LIB_PATH=$R/exp_libs/$v.so PYTHONPATH=$R/scripts/ab/hook:$PYTHONPATH; fi
  rm -rf /tmp/pf_dw
  rocprofv3 --kernel-trace -d /tmp/pf_dw -- python $R/bench.py --legs none --steps 6 --warmup 2 > /tmp/dw_$v.json 2>/dev/null
  DB=$(find /tmp/pf_dw -name "*.db" | head -1)
  echo "== $v $(python -c "import json;print(json.loads(open('/tmp/dw_$v.json').read().strip().splitlines()[-1])['ms_per_step'])")"
  python $R/scripts/prof_summary.py $DB | grep -E "k_dw3x3" | cut -c1-150
done

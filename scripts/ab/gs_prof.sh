# per-kernel averages of the rvs1024 leg for library variants: bash scripts/ab/gs_prof.sh [tag ...]
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in ${@:-tree}; do
  if [ "$v" = tree ]; then unset AADG_LIB_PATH; else export AADG_LIB_PATH=$R/exp_libs/$v.so PYTHONPATH=$R/scripts/ab/hook:$PYTHONPATH; fi
  rm -rf /tmp/prof_rvs
  rocprofv3 --kernel-trace --stats -d /tmp/prof_rvs -- python $R/bench.py --only_legs rvs1024 > /dev/null 2>&1
  DB=$(find /tmp/prof_rvs -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB /tmp/rvs_$v.txt > /dev/null
  echo "== $v"; grep -E "^kernel|^k_" /tmp/rvs_$v.txt | cut -c1-150
done

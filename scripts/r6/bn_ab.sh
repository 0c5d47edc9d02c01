cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in old new; do
  rm -rf /tmp/prof_bn
  if [ $v = old ]; then export AADG_LIB_PATH=$R/exp_libs/morning.so PYTHONPATH=$R/scripts/ab/hook; else unset AADG_LIB_PATH; unset PYTHONPATH; fi
  rocprofv3 --kernel-trace --stats -d /tmp/prof_bn -- python $R/scripts/r6/bn_time.py > /dev/null 2>&1
  DB=$(find /tmp/prof_bn -name "*.db" | head -1); python $R/scripts/prof_summary.py $DB /tmp/bn_stats.txt > /dev/null
  echo "== $v"; grep "^k_bn_reduce" /tmp/bn_stats.txt | cut -c1-120
done

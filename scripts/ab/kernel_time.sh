# rocprofv3 average of kernels matching $1 in the aug512 / rvs1024 legs for library variants: bash scripts/ab/kernel_time.sh <kernel> [tag ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
K=$1; shift
for v in ${@:-tree}; do
  if [ "$v" = tree ]; then unset AADG_LIB_PATH; else export AADG_LIB_PATH=$R/exp_libs/$v.so PYTHONPATH=$R/scripts/ab/hook:$PYTHONPATH; fi
  for leg in aug512 rvs1024; do
    rm -rf /tmp/prof_kt
    rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -- python $R/bench.py --only_legs $leg > /dev/null 2>&1
    DB=$(find /tmp/prof_kt -name "*.db" | head -1)
    echo "$v $leg $(python $R/scripts/prof_summary.py $DB | grep "^$K" | cut -c1-130)"
  done
done

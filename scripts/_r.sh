cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/scripts/quick_time_rvs.py 1024 1024
rocprofv3 --kernel-trace --stats -d /tmp/prof_r -- python $R/scripts/quick_time_rvs.py 1024 1024 > /dev/null 2>&1
python $R/scripts/prof_summary.py $(find /tmp/prof_r -name "*.db" | head -1) | head -8 | cut -c1-150

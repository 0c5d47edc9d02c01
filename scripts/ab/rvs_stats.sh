# per-kernel durations of the rvs1024 leg (rocprofv3 --kernel-trace --stats) for library variants exp_libs/<tag>.so -> gpurun_out/exp/rvs_stats_<tag>.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/exp
for v in "$@"; do
  cp $R/exp_libs/$v.so $R/aadg_amd/lib/libaadg_hip.so
  rm -rf /tmp/prof_rvs_q
  rocprofv3 --kernel-trace --stats -d /tmp/prof_rvs_q -- python $R/bench.py --only_legs rvs1024 > /dev/null 2>&1
  DB=$(find /tmp/prof_rvs_q -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB $R/gpurun_out/exp/rvs_stats_$v.txt > /dev/null
  echo "== $v"; head -7 $R/gpurun_out/exp/rvs_stats_$v.txt | cut -c1-175
done

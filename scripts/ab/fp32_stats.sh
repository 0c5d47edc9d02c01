# per-kernel durations of the float32-backbone step -> gpurun_out/exp/fp32_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/exp
rm -rf /tmp/prof_fp32
rocprofv3 --kernel-trace --stats -d /tmp/prof_fp32 -- python $R/bench.py --legs none --backbone_dtype fp32 --steps 4 --warmup 2 > /dev/null 2>&1
DB=$(find /tmp/prof_fp32 -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB $R/gpurun_out/exp/fp32_stats.txt > /dev/null
head -40 $R/gpurun_out/exp/fp32_stats.txt | cut -c1-160

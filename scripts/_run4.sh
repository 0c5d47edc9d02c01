mkdir -p gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d /tmp/prof_s8 -- python $R/bench.py --legs none --shard_of 8 --steps 30 --warmup 5 > $R/gpurun_out/r2/shard8_prof.json 2>/dev/null
DB=$(find /tmp/prof_s8 -name "*.db" | head -1)
python $R/scripts/busy_summary.py $DB "k_stem7x7<" 8 20 | tee $R/gpurun_out/r2/shard8_busy.txt
rocprofv3 --kernel-trace -d /tmp/prof_s1 -- python $R/bench.py --legs none --steps 12 --warmup 3 > /dev/null 2>&1
DB=$(find /tmp/prof_s1 -name "*.db" | head -1)
python $R/scripts/busy_summary.py $DB "k_stem7x7<" 4 8 | tee $R/gpurun_out/r2/n1_busy.txt

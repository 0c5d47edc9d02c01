mkdir -p gpurun_out/r2
bash scripts/make_profiles.sh > gpurun_out/r2/make_profiles.log 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --legs none --shard_of 8 --steps 30 --warmup 5 > $R/gpurun_out/r2/shard8.json 2>$R/gpurun_out/r2/shard8.err
rocprofv3 --kernel-trace --stats -d /tmp/prof_s8 -- python $R/bench.py --legs none --shard_of 8 --steps 30 --warmup 5 > $R/gpurun_out/r2/shard8_prof.json 2>/dev/null
DB=$(find /tmp/prof_s8 -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB $R/gpurun_out/r2/shard8_kernel_stats.txt > /dev/null
python $R/scripts/busy_summary.py $DB 0.5 > $R/gpurun_out/r2/shard8_busy.txt
cat $R/gpurun_out/r2/shard8_busy.txt
python -c "import json;d=json.loads(open('$R/gpurun_out/r2/shard8.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['value'])"

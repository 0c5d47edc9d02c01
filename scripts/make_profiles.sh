# Regenerates the round-2 profile artefacts on a GPU box (run through gpurun; results land in gpurun_out/r2/profiles/, copy the
# ones to keep into profiles/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2/profiles
mkdir -p $O
# 1. per-kernel durations of the bench command
rocprofv3 --kernel-trace --stats -d /tmp/prof_b -- python $R/bench.py --legs none --steps 10 --warmup 3 > $O/r02_bench_under_rocprof.json 2>/dev/null
DB=$(find /tmp/prof_b -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB $O/r02_bench_rocprofv3_kernel_stats.txt > /dev/null
# 2. HBM traffic of the dominant kernel (separate passes)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pmc_$c -- python $R/bench.py --legs none --steps 5 --warmup 2 > /dev/null 2>&1
  python $R/scripts/pmc_summary.py /tmp/pmc_$c "k_fused3" > $O/r02_pmc_$c.txt
done
# 3. the bench line itself (all legs), not under a profiler
python $R/bench.py > $O/r02_bench_n1.json 2> $O/r02_bench_n1.err
grep -n "k_fused3\|k_hist\|k_lut\|k_tables\|k_sinkhorn\|k_embed\|k_seg" $O/r02_bench_rocprofv3_kernel_stats.txt
cat $O/r02_pmc_FETCH_SIZE.txt $O/r02_pmc_WRITE_SIZE.txt

# Round-3 profile artefacts (run through gpurun; results land in gpurun_out/r3/<tag>/, copy the ones to keep into profiles/).
#   bash scripts/make_profiles_r03.sh <tag> [parts]      parts: any of  fop rvs bench shard8  (default: fop rvs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-final}
PARTS=${2:-"fop rvs"}
O=$R/gpurun_out/r3/$TAG
mkdir -p $O

for part in $PARTS; do
case $part in
fop)
  # float tensor ops: the leg itself, per-kernel durations, HBM traffic per kernel (separate passes)
  python $R/bench.py --only_legs fop,kernels > $O/fop_leg.json 2> $O/fop_leg.err
  rocprofv3 --kernel-trace --stats -d /tmp/prof_fop -- python $R/bench.py --only_legs fop,kernels > $O/fop_leg_under_rocprof.json 2>/dev/null
  DB=$(find /tmp/prof_fop -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB $O/fop_kernel_stats.txt > /dev/null
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d /tmp/pmc_fop_$c -- python $R/bench.py --only_legs fop > /dev/null 2>&1
    python $R/scripts/pmc_by_kernel.py /tmp/pmc_fop_$c > $O/pmc_fop_$c.txt
  done
  python $R/scripts/fop_traffic.py $O/fop_leg.json $O/pmc_fop_FETCH_SIZE.txt $O/pmc_fop_WRITE_SIZE.txt > $O/fop_traffic.json
  ;;
rvs)
  # BASELINE configs[2]: RVS pipeline at 1024 x 1024, the tile kernels' durations, traffic and issue counters
  python $R/bench.py --only_legs rvs1024 > $O/rvs1024_leg.json 2> $O/rvs1024_leg.err
  rocprofv3 --kernel-trace --stats -d /tmp/prof_rvs -- python $R/bench.py --only_legs rvs1024 > /dev/null 2>&1
  DB=$(find /tmp/prof_rvs -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB $O/rvs1024_kernel_stats.txt > /dev/null
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d /tmp/pmc_rvs_$c -- python $R/bench.py --only_legs rvs1024 > /dev/null 2>&1
    python $R/scripts/pmc_by_kernel.py /tmp/pmc_rvs_$c > $O/pmc_rvs1024_$c.txt
  done
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d /tmp/pmc_rvs_sq$i -- python $R/bench.py --only_legs rvs1024 > /dev/null 2>&1
    python $R/scripts/pmc_by_kernel.py /tmp/pmc_rvs_sq$i "k_" > $O/pmc_rvs1024_sq$i.txt
  done
  cat $O/pmc_rvs1024_sq*.txt > $O/rvs1024_pmc.txt
  ;;
bench)
  rocprofv3 --kernel-trace --stats -d /tmp/prof_b -- python $R/bench.py --legs none --steps 10 --warmup 3 > $O/bench_under_rocprof.json 2>/dev/null
  DB=$(find /tmp/prof_b -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB $O/bench_rocprofv3_kernel_stats.txt > /dev/null
  python $R/scripts/busy_summary.py $DB "k_stem7x7<" 3 8 > $O/bench_busy.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d /tmp/pmc_b_$c -- python $R/bench.py --legs none --steps 5 --warmup 2 > /tmp/bench_$c.json 2>/dev/null
    python $R/scripts/pmc_by_kernel.py /tmp/pmc_b_$c "k_fused3" > $O/pmc_bench_$c.txt
  done
  python $R/bench.py > $O/bench_n1.json 2> $O/bench_n1.err
  ;;
shard8)
  rocprofv3 --kernel-trace --stats -d /tmp/prof_s8 -- python $R/bench.py --legs none --shard_of 8 --steps 30 --warmup 5 > $O/shard8_under_rocprof.json 2>/dev/null
  DB8=$(find /tmp/prof_s8 -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB8 $O/shard8_rocprofv3_kernel_stats.txt > /dev/null
  python $R/scripts/busy_summary.py $DB8 "k_stem7x7<" 8 20 > $O/shard8_busy.txt
  python $R/bench.py --legs none --shard_of 8 --steps 30 --warmup 5 > $O/shard8.json 2>/dev/null
  ;;
esac
done
ls -la $O

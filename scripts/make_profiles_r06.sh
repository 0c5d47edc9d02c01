# Round-6 profile artefacts (run through gpurun; results land in gpurun_out/r6p/<tag>/, copy the ones to keep into profiles/).
#   bash scripts/make_profiles_r06.sh <tag> [parts]      parts: any of  fop aug512 rvs bench shard8 segformer x3layers fp32lib bf16 skbig ctrl dist hosthot segloss  (default: fop aug512 rvs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-final}
PARTS=${2:-"fop aug512 rvs"}
O=$R/gpurun_out/r6p/$TAG
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
aug512)
  # BASELINE configs[1]'s augmentation call on its own (the seeded hot-path batches, no backbone around it): per-kernel durations
  python $R/bench.py --only_legs aug512 > $O/aug512_leg.json 2> $O/aug512_leg.err
  rocprofv3 --kernel-trace --stats -d /tmp/prof_a5 -- python $R/bench.py --only_legs aug512 > /dev/null 2>&1
  DB=$(find /tmp/prof_a5 -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB $O/aug512_kernel_stats.txt > /dev/null
  # the statistics pass by late-unit class (scripts/ab/stat_classes.py)
  bash $R/scripts/ab/stat_classes.sh > $O/stat_pass_classes.txt 2>&1
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
  python $R/scripts/rvs_traffic.py $O/rvs1024_leg.json $O/pmc_rvs1024_FETCH_SIZE.txt $O/pmc_rvs1024_WRITE_SIZE.txt $O/rvs1024_kernel_stats.txt > $O/rvs1024_traffic.json
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
  # the derived JSON files bench.py quotes (kernel averages of the profiled run, HBM traffic of k_fused3 per unit) -- written BEFORE the
  # final un-profiled bench run, which reads them
  python - <<PY
import json, re
O = "$O"
def avg(name):
    n, tot = 0, 0.0                      # every instantiation of a template kernel (k_hist_fused<false> / <true>)
    for l in open(O + "/bench_rocprofv3_kernel_stats.txt"):
        if l.startswith(name + " ") or l.startswith(name + "<"):
            p = l[len(l.split()[0]):].split()
            n += int(p[0]); tot += int(p[0]) * float(p[1]) / 1e3
    return n, (tot / n if n else 0.0)
stats = {"file": "r06_bench_rocprofv3_kernel_stats.txt",
         "command": "rocprofv3 --kernel-trace --stats -- python bench.py --legs none --steps 10 --warmup 3"}
nf = avg("k_fused3")[0]
for k in ("k_fused3", "k_luts_tables", "k_hist_fused", "k_lut"):
    n, a = avg(k)
    stats[k + "_avg_ms"] = round(a, 4)
    stats[k + "_calls_per_launch"] = round(n / max(nf, 1), 2)
json.dump(stats, open(O + "/bench_kernel_stats.json", "w"))
def pmc(c):
    t = open(O + "/pmc_bench_%s.txt" % c).read()
    return float(re.search(r"avg\s+([\d.]+)", t).group(1))
b = json.load(open("/tmp/bench_FETCH_SIZE.json"))["roofline"]
f, w, units = pmc("FETCH_SIZE"), pmc("WRITE_SIZE"), b["units_per_launch"]
hbm = int(2 * f * 1024 + w * 1024)
json.dump({"kernel": "k_fused3", "file": "r06_traffic_k_fused3.json", "size": 512, "units_measured": units,
           "command": "rocprofv3 --pmc FETCH_SIZE -- python bench.py --legs none --steps 5 --warmup 2   (and a second, separate pass with --pmc WRITE_SIZE); scripts/make_profiles_r06.sh",
           "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w,
           "correction": "gfx950 FETCH_SIZE counts 64 B per 128 B request: doubled (MI355X_MICROARCH.md, HBM section); WRITE_SIZE used as is",
           "hbm_bytes_per_launch": hbm, "hbm_bytes_per_unit": hbm // units, "kernel_bytes": b["bytes_per_launch"],
           "note": "traffic / bytes the kernel addresses = %.3f; bench.py scales hbm_bytes_per_unit by the units of its launch" % (hbm / b["bytes_per_launch"])},
          open(O + "/traffic_k_fused3.json", "w"), indent=2)
PY
  # the committed copies bench.py reads live in profiles/: refresh them so that the line below quotes THIS run's profile
  cp $O/bench_kernel_stats.json $R/profiles/r06_bench_kernel_stats.json
  cp $O/traffic_k_fused3.json $R/profiles/r06_traffic_k_fused3.json
  if [ -f $O/rvs1024_traffic.json ]; then cp $O/rvs1024_traffic.json $R/profiles/r06_rvs1024_traffic.json; fi
  python $R/bench.py --detail $O/bench_detail.json > $O/bench_n1.json 2> $O/bench_n1.err
  ;;
segformer)
  # BASELINE configs[4] (SegFormer-B2, 8 domains, bf16): the 48 rows of one of 8 ranks on one GPU
  SF="--legs none --cfg experiments/merged_sinkhorn/segformer_b2_d8.yaml --shard_of 8"
  rocprofv3 --kernel-trace --stats -d /tmp/prof_sf -- python $R/bench.py $SF --steps 9 --warmup 2 > /dev/null 2>&1
  DBS=$(find /tmp/prof_sf -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DBS $O/segformer_48rows_kernel_stats.txt > /dev/null
  python $R/scripts/busy_summary.py $DBS "k_upsample_sum_plane" 3 6 > $O/segformer_48rows_busy.txt
  python $R/bench.py $SF --steps 20 --warmup 3 > $O/segformer_48rows.json 2>/dev/null
  ;;
shard8)
  rocprofv3 --kernel-trace --stats -d /tmp/prof_s8 -- python $R/bench.py --legs none --shard_of 8 --steps 30 --warmup 5 > $O/shard8_under_rocprof.json 2>/dev/null
  DB8=$(find /tmp/prof_s8 -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB8 $O/shard8_rocprofv3_kernel_stats.txt > /dev/null
  python $R/scripts/busy_summary.py $DB8 "k_stem7x7<" 8 20 > $O/shard8_busy.txt
  python $R/bench.py --legs none --shard_of 8 --steps 30 --warmup 5 > $O/shard8.json 2>/dev/null
  ;;
x3layers)
  # per-layer times of the float32-precision (f32x3) convolution kernels on the backbone's shapes, with both rooflines
  python $R/scripts/x3_layer_times.py 144 > $O/x3_layer_times.txt 2>&1
  ;;
fp32lib)
  # the step on the library's float32 convolutions (what f32x3 replaces): bench line + kernel table
  python $R/bench.py --legs none --backbone_dtype fp32 --steps 10 --warmup 2 > $O/fp32_library_n1.json 2>/dev/null
  rocprofv3 --kernel-trace --stats -d /tmp/prof_f32 -- python $R/bench.py --legs none --backbone_dtype fp32 --steps 4 --warmup 2 > /dev/null 2>&1
  DBF=$(find /tmp/prof_f32 -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DBF $O/fp32_library_kernel_stats.txt > /dev/null
  ;;
bf16)
  # the step under bfloat16 autocast (secondary figure): kernel table
  rocprofv3 --kernel-trace --stats -d /tmp/prof_b16 -- python $R/bench.py --legs none --backbone_dtype bf16 --steps 10 --warmup 3 > $O/bf16_under_rocprof.json 2>/dev/null
  DBB=$(find /tmp/prof_b16 -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DBB $O/bf16_kernel_stats.txt > /dev/null
  ;;
ctrl)
  # the fused controller calls: event times of sample / 5-epoch PPO update, kernel table (k_ctrl_sample_seq, k_ppo_rollout, k_ppo_grad_adam)
  python $R/scripts/ubench/ctrl_time.py 300 > $O/controller_times.txt 2>/dev/null
  bash $R/scripts/ubench/ctrl_prof.sh > $O/controller_kernel_stats.txt 2>/dev/null
  ;;
dist)
  # round 6: the plain step against the ONE rank run through the whole distributed path over RCCL (own gradient reducer, weight-gradient
  # stream on, synchronised BatchNorm with the on-load fusions), three interleaved pairs on this box; then its kernel table
  (cd $R && bash scripts/r6/dist_ab.sh 3 > /dev/null 2>&1; cp gpurun_out/r6/dist_ab.txt $O/dist_ab.txt)
  rocprofv3 --kernel-trace --stats -d /tmp/prof_fd -- python $R/bench.py --legs none --steps 8 --warmup 3 --force_dist --dist_backend nccl > $O/force_dist_under_rocprof.json 2>/dev/null
  DBD=$(find /tmp/prof_fd -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DBD $O/force_dist_kernel_stats.txt > /dev/null
  ;;
hosthot)
  python $R/scripts/hot_host_profile.py > $O/hot_host_profile.txt 2>&1
  ;;
segloss)
  python $R/scripts/r6/segloss_time.py > $O/segloss_time.txt 2>&1
  python $R/scripts/r6/dw_time.py > $O/dw_time.txt 2>&1
  ;;
skbig)
  # SURVEY 8(d)'s scaled synthetic Sinkhorn (3 x 4096^2 points): wall time and kernel table
  python $R/scripts/quick_time_sinkhorn_big.py 4096 > $O/sinkhorn_big_time.txt 2>&1
  rocprofv3 --kernel-trace --stats -d /tmp/prof_skb -- python $R/scripts/quick_time_sinkhorn_big.py 4096 > /dev/null 2>&1
  DBK=$(find /tmp/prof_skb -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DBK $O/sinkhorn_big_kernel_stats.txt > /dev/null
  ;;
esac
done
ls -la $O

# Round 5: kernel table of the reference-precision (float32 backbone) step.   bash scripts/prof_fp32_r05.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-before}
O=$R/gpurun_out/r5/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats -d /tmp/prof_f32 -- python $R/bench.py --legs none --backbone_dtype fp32 --steps 4 --warmup 2 > $O/fp32_under_rocprof.json 2> $O/fp32.err
DB=$(find /tmp/prof_f32 -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB $O/fp32_kernel_stats.txt > /dev/null
head -60 $O/fp32_kernel_stats.txt

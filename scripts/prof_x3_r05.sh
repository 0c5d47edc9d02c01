# Round 5: the float32-precision (f32x3) step: bench line with the precision leg, then its kernel table.   bash scripts/prof_x3_r05.sh <tag> [dtype]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x3}
DT=${2:-f32x3}
O=$R/gpurun_out/r5/$TAG
mkdir -p $O
python $R/bench.py --backbone_dtype $DT --legs precision --steps 8 --warmup 3 > $O/bench_$DT.json 2> $O/bench_$DT.err
tail -c 3000 $O/bench_$DT.json
rocprofv3 --kernel-trace --stats -d /tmp/prof_x3 -- python $R/bench.py --legs none --backbone_dtype $DT --steps 4 --warmup 2 > $O/${DT}_under_rocprof.json 2> $O/${DT}_rocprof.err
DB=$(find /tmp/prof_x3 -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB $O/${DT}_kernel_stats.txt > /dev/null
head -50 $O/${DT}_kernel_stats.txt | cut -c1-140

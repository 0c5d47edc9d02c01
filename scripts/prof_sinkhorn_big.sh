# large-cloud Sinkhorn (SURVEY 8d's scaled synthetic): tests, wall time, kernel table.   bash scripts/prof_sinkhorn_big.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5/${1:-skb}
mkdir -p $O
cd $R && python -m pytest tests/test_gpu_sinkhorn.py -q -x 2>&1 | tail -5
cd /tmp
python $R/scripts/quick_time_sinkhorn_big.py 4096 2>&1 | tee $O/time.txt
rocprofv3 --kernel-trace --stats -d /tmp/prof_skb -- python $R/scripts/quick_time_sinkhorn_big.py 4096 > /dev/null 2>&1
DB=$(find /tmp/prof_skb -name "*.db" | head -1)
python $R/scripts/prof_summary.py $DB $O/kernel_stats.txt | grep "k_big" | cut -c1-140

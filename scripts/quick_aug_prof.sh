# quick per-kernel view of the augmentation call at 512 (configs[1]) and 1024 (configs[2]); results in gpurun_out/q/<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-q}
O=$R/gpurun_out/q/$TAG
mkdir -p $O
for leg in aug512 rvs1024; do
  python $R/bench.py --only_legs $leg > $O/$leg.json 2> $O/$leg.err
  rm -rf /tmp/prof_$leg
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$leg -- python $R/bench.py --only_legs $leg > /dev/null 2>&1
  DB=$(find /tmp/prof_$leg -name "*.db" | head -1)
  python $R/scripts/prof_summary.py $DB $O/${leg}_kernel_stats.txt > /dev/null
done
python - <<PY
import json
for leg,key in (("aug512","aug_512"),("rvs1024","rvs_1024")):
    d=json.load(open("$O/%s.json"%leg))[key]["roofline"]
    print(leg, "tile kernels %.4f ms frac %.3f | call %.4f ms frac %.3f" % (d["kernel_ms"], d["frac"], d["stage"]["ms"], d["stage"]["frac"]))
PY
head -12 $O/aug512_kernel_stats.txt | cut -c1-150
head -12 $O/rvs1024_kernel_stats.txt | cut -c1-150

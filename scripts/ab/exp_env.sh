# rvs1024 leg with the current library under different environment settings: bash scripts/ab/exp_env.sh VAR v1 v2 ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --only_legs rvs1024 > gpurun_out/exp/rvs_env_$v.json 2> gpurun_out/exp/rvs_env_$v.err
  echo -n "$VAR=$v  "; python scripts/show_rvs.py gpurun_out/exp/rvs_env_$v.json
done

cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "morning: "; AADG_LIB_PATH=exp_libs/morning.so PYTHONPATH=scripts/ab/hook python bench.py --legs none --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'])"
  echo -n "tree:    "; python bench.py --legs none --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'])"
done

# whole-step A/B of a host-side switch: bash scripts/r6/ab_exec.sh "<python statement that restores the OLD behaviour>"   (three interleaved pairs)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "old:  "; AADG_AB_EXEC="$1" PYTHONPATH=scripts/ab/hook python bench.py --legs none --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'])"
  echo -n "tree: "; python bench.py --legs none --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'])"
done

cd $GRAFT_REPO_ROOT
X='import torch, aadg_amd.models.deeplab as d; d._INPLACE_CONCAT_DTYPES = (torch.bfloat16,)'
for i in 1 2 3; do
  echo -n "torch.cat: "; AADG_AB_EXEC="$X" PYTHONPATH=scripts/ab/hook python bench.py --legs none --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'])"
  echo -n "in place:  "; python bench.py --legs none --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(b['ms_per_step'])"
done

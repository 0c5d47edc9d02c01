# statistics pass by late-unit class for library variants: bash scripts/ab/stat_classes.sh [tag ...]   (tag = exp_libs/<tag>.so; none = the tree's)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${@:-tree}; do
  rm -rf /tmp/prof_sc
  if [ "$v" = tree ]; then unset AADG_LIB_PATH; else export AADG_LIB_PATH=$R/exp_libs/$v.so PYTHONPATH=$R/scripts/ab/hook:$PYTHONPATH; fi
  STAT_CLASSES_OUT=/tmp/stat_classes.json rocprofv3 --kernel-trace -d /tmp/prof_sc -- python $R/scripts/ab/stat_classes.py > /tmp/sc.log 2>&1 || tail -20 /tmp/sc.log
  DB=$(find /tmp/prof_sc -name "*.db" | head -1)
  echo "== $v"
  python $R/scripts/ab/stat_classes_read.py $DB /tmp/stat_classes.json
done

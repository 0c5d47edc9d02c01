# rocprofv3 kernel averages of the fused controller calls (sample + PPO update) for the built library: bash scripts/ubench/ctrl_prof.sh
case "$AADG_LIB_PATH" in /*|"") ;; *) export AADG_LIB_PATH=$GRAFT_REPO_ROOT/$AADG_LIB_PATH;; esac
[ -n "$AADG_LIB_PATH" ] && export PYTHONPATH=$GRAFT_REPO_ROOT/scripts/ab/hook:$PYTHONPATH
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pc
rocprofv3 --kernel-trace --stats -d /tmp/pc -- python $GRAFT_REPO_ROOT/scripts/ubench/ctrl_time.py 100 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/scripts/prof_summary.py $(find /tmp/pc -name "*.db" | head -1) /tmp/pc/stats.txt > /dev/null
grep -E "^kernel|k_ppo|k_ctrl" /tmp/pc/stats.txt

# round 5, call 17: the suite once more with Tracker.__del__ (no tracker left open: the level form is what the tracker tests run)
cd $GRAFT_REPO_ROOT
echo "=== final tree, Tracker.__del__ in place: python -m pytest tests/ -x -q -m gpu -rs" > gpurun_out/r05_suite_repeats_final3.log
python -m pytest tests/ -x -q -m gpu -rs 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | grep -E "passed|failed|SKIPPED|^FAILED|^E " | tail -5 >> gpurun_out/r05_suite_repeats_final3.log
cat gpurun_out/r05_suite_repeats_final3.log

# round 5, call 10: the whole GPU suite five times back to back -> profiles/r05_suite_repeats.log
cd $GRAFT_REPO_ROOT
: > gpurun_out/r05_suite_repeats.log
for i in 1 2 3 4 5; do
  echo "=== run $i: python -m pytest tests/ -x -q -m gpu" >> gpurun_out/r05_suite_repeats.log
  python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -4 >> gpurun_out/r05_suite_repeats.log
done
cat gpurun_out/r05_suite_repeats.log
python -c "import __graft_entry__ as g; g.smoke()"

# r04 call 2: the whole GPU suite on the tree (lean voxel kernel by default, ThreadObject shell, plan hooks), the extended issue-rate table
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### full GPU suite"
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/c2_tests.log 2>&1; grep -E "passed|failed|error" gpurun_out/c2_tests.log | tail -3; grep -E "^(FAILED|ERROR)|Error|assert " gpurun_out/c2_tests.log | head -20
echo "#### valu rates (extended)"
python scripts/valu_rates.py > gpurun_out/r04_valu_rates.md 2> gpurun_out/r04_valu_rates.err; tail -22 gpurun_out/r04_valu_rates.md; tail -2 gpurun_out/r04_valu_rates.err
echo "#### bench default"
python bench.py 2>/dev/null | tail -1 > gpurun_out/c2_bench.json; cut -c1-1500 gpurun_out/c2_bench.json

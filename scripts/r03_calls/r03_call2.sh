cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### variants"; KT_SKIP_TREE=1 bash scripts/variants_ab.sh 40 2>&1 | tee gpurun_out/r03_variants_call2.log
echo "#### pmc"; bash scripts/pmc_variants.sh 2>&1 | grep PMC | tee gpurun_out/r03_pmc_variants_call2.log

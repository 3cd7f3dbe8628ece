cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
echo "#### pytest -m gpu (tree)"; python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "#### div check"; python - <<'PY'
import ctypes as C
from kintinuous_amd import abi
ctx = abi.Ctx(0); out = (C.c_uint * 3)()
abi._chk(abi.lib().kt_debug_div_check(ctx.h, out)); print("markstein mismatches", out[0], "worst |n| bits", hex(out[1]), "mismatches >= 2^-100", out[2])
PY
echo "#### valu rates"; python scripts/valu_rates.py | tee gpurun_out/r03_valu_rates.md
echo "#### variants"; bash scripts/variants_ab.sh 40 2>&1 | tee gpurun_out/r03_variants_call1.log
echo "#### calibration"; bash scripts/pmc_calibrate.sh 2>&1 | tee gpurun_out/r03_pmc_calibration.log

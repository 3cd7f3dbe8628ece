"""Issue cost of the instruction kinds tsdf23 is made of (kt_debug_valu_rates, csrc/kt_debug.hip), in shader cycles per
wave-instruction per SIMD at 1 / 2 / 4 / 8 resident waves per SIMD.   python scripts/valu_rates.py > gpurun_out/r03_valu_rates.md"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kintinuous_amd import abi

KINDS = ["v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "v_rndne_f32 / v_cvt_i32_f32", "v_fma_f32 + s_add_u32 (per pair)", "4 v_readlane + 4 v_fma (per VALU)",
         "v_sqrt_f32", "v_cmp_gt_f32 / v_cndmask_b32", "v_pk_add_f32", "v_mad_u32_u24", "v_cvt_f32_ubyte0"]
ctx = abi.Ctx(0)
out = (C.c_double * 4)()
print("| instruction | " + " | ".join("%d waves/SIMD" % w for w in (1, 2, 4, 8)) + " |")
print("|---|---|---|---|---|")
for kind, name in enumerate(KINDS):
    cells = []
    for w in (1, 2, 4, 8):
        abi._chk(abi.lib().kt_debug_valu_rates(ctx.h, kind, 2000, w, out))
        # ns of wall clock the SIMD spends per wave-instruction with w waves interleaved (and s_memtime ticks per instruction per wave)
        cells.append("%.2f ns (%.1f ticks/wave)" % (out[3] * 1e6 / (w * out[2]), out[0] / out[2]))
    print("| `%s` | " % name + " | ".join(cells) + " |")
print("\n(launch duration / (waves per SIMD x 64 000 instructions): at 2.4 GHz a 2-cycle wave64 instruction is 0.83 ns)")

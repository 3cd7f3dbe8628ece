"""Issue cost of the instruction kinds tsdf23 is made of (kt_debug_valu_rates, csrc/kt_debug.hip) at 1 / 2 / 4 / 8 resident waves per
SIMD, with the shader clock MEASURED under each load (s_memtime ticks against the constant 100 MHz s_memrealtime), so that cycles per
wave-instruction do not rest on the data sheet's 2.4 GHz.   python scripts/valu_rates.py > gpurun_out/r04_valu_rates.md"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kintinuous_amd import abi

KINDS = ["v_fma_f32", "v_pk_fma_f32", "v_rcp_f32", "v_rndne_f32 / v_cvt_i32_f32", "v_fma_f32 + s_add_u32 (1 : 1)", "4 v_readlane + 4 v_fma",
         "v_sqrt_f32", "v_cmp_gt_f32 / v_cndmask_b32", "v_pk_add_f32", "v_mad_u32_u24", "v_cvt_f32_ubyte0",
         "s_add_u32 only", "2 v_fma_f32 : 1 s_add_u32", "1 v_fma_f32 : 2 s_add_u32", "v_cmp -> SGPR pair + s_and_b64 (1 : 1)",
         "s_and_saveexec / v_fma / s_or exec (1 : 2)"] + \
        ['v_add_f32', 'v_mul_f32', 'v_max_f32', 'v_add_u32', 'v_and_b32', 'v_lshlrev_b32', 'v_mov_b32', 'v_cndmask_b32 (vcc fixed)', 'v_cmp_gt_f32 only', 'v_cvt_f32_u32', 'v_med3_i32', 'v_mul_u32_u24', 'v_lshl_add_u32', 'v_mad_u64_u32 (pair dst)', 'v_bfe_u32', 'v_min_u32', 'v_sub_f32 |abs| (VOP3)', 'v_cmp_lt_u32 to SGPR pair (VOP3)'] + \
        ['v_sub_u32', 'v_or_b32', 'v_xor_b32', 'v_min_f32', 'v_cvt_i32_f32', 'v_rndne_f32', 'v_perm_b32', 'v_bfi_b32', 'v_and_or_b32', 'v_or3_b32', 'v_add3_u32', 'v_lshl_or_b32', 'v_floor_f32', 'v_cvt_u32_f32', 'v_subrev_f32', 'v_fma_f32 with SGPR operand']
ctx = abi.Ctx(0)
out = (C.c_double * 8)()
print("# Issue cost per wave-instruction and SIMD, shader clock measured under the load (round 4)\n")
print("cell = ns of wall clock per wave-instruction per SIMD | shader cycles per wave-instruction per SIMD at the MEASURED clock | that clock in MHz\n")
print("| instruction stream | " + " | ".join("%d waves/SIMD" % w for w in (1, 2, 4, 8)) + " |")
print("|---|---|---|---|---|")
for kind, name in enumerate(KINDS):
    cells = []
    for w in (1, 2, 4, 8):
        abi._chk(abi.measure_lib().kt_debug_valu_rates(ctx.h, kind, 2000, w, out))
        span_us, mhz, n = out[5], out[4], out[2]
        ns = span_us * 1e3 / (w * n)          # first wave in .. last wave out, per instruction of the SIMD's w waves
        cells.append("%.2f ns, %.2f cyc, %.0f MHz" % (ns, ns * mhz * 1e-3, mhz))
    print("| `%s` | " % name + " | ".join(cells) + " |")
print("\n(span = first wave in .. last wave out in s_memrealtime; cycles = ns x measured MHz; a stream of V VALU and S SALU per body is counted as V + S instructions)")

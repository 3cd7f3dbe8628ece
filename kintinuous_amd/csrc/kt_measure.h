/* kt_measure.h -- libkt_debug.so: measurement kernels that are NOT in the product library (round 6).  Loaded by
 * scripts/pmc_calibrate.py, scripts/valu_rates.py and tests/test_gpu_volume.py next to libkt_hip.so, whose context type and error reporting it uses. */
#ifndef KT_MEASURE_H
#define KT_MEASURE_H

#include "../../include/kt_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* PMC calibration hook: stream `bytes` of a device buffer with 2- or 4-byte-per-lane coalesced accesses (the widths of the tsdf /
 * colour volume accesses); rmw = 0 reads, 1 reads and writes back.  Used by scripts/pmc_calibrate.py to scale FETCH_SIZE / WRITE_SIZE. */
int kt_debug_stream(kt_ctx* ctx, void* buf, size_t bytes, int elem_size, int rmw);
/* PMC calibration on the voxel kernel's own access pattern: a wave owns a 32 x 2 wave-column of an N x N x Z array and walks z, so an
 * access is two 32-lane rows (64 B at elem_size 2, 128 B at 4).  halves = 2 reads every element once; halves = 1 only the even
 * wave-columns (elem_size 2: the left 64 bytes of every 128-byte line).  N % 32 == 0, Z % 4 == 0. */
int kt_debug_stream_rows(kt_ctx* ctx, void* buf, int N, int Z, int elem_size, int halves, int rmw);
/* issue cost of one instruction kind (csrc/kt_debug.hip lists them) at waves_per_simd resident waves: out_host = {mean, max shader
 * ticks per wave for the loop, wave-instructions per wave, launch duration in ms, shader clock in MHz while the loop ran (s_memtime
 * against the 100 MHz s_memrealtime), first wave in .. last wave out in us, VALU per wave, SALU per wave} */
int kt_debug_valu_rates(kt_ctx* ctx, int kind, int iters, int waves_per_simd, double out_host[8]);
/* test hook: the voxel kernel's division shortcut (table reciprocal + one correction) against the IEEE division for every finite float
 * numerator and every divisor 1..256: out_host = {mismatches, float bits of the largest |numerator| among them, mismatches at |n| >= 2^-100} */
int kt_debug_div_check(kt_ctx* ctx, unsigned int out_host[3]);

#ifdef __cplusplus
}
#endif
#endif

"""Launches the calibration streams (known byte counts); run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (scripts/pmc_calibrate.sh).
1. contiguous streams at tsdf23's access widths (kt_debug_stream: 64 lanes x 2 B / 4 B per access);
2. tsdf23's REAL pattern (kt_debug_stream_rows): a wave owns a 32 x 2 wave-column of a 768^3 array and walks z -- two 32-lane rows per
   access (64 B of tsdf / 128 B of colour each), both halves of a line read by x-neighbouring waves, or only the left halves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kintinuous_amd import abi
ctx = abi.Ctx(0)
GB = 1 << 30
N = 768
buf = abi.DevBuf(ctx, max(2 * GB, N * N * N * 4))
abi._chk(abi.lib().kt_memset(ctx.h, buf.ptr, 1, buf.nbytes))
ctx.sync()
for elem, rmw in ((2, 0), (4, 0), (2, 1), (4, 1)):
    for rep in range(3):
        abi._chk(abi.measure_lib().kt_debug_stream(ctx.h, buf.ptr, 2 * GB, elem, rmw))
    ctx.sync()
print("streamed 2 GiB per launch: u16 read, u32 read, u16 rmw, u32 rmw (3 launches each)")
for elem, halves, rmw in ((2, 2, 0), (2, 1, 0), (4, 2, 0), (4, 1, 0), (2, 2, 1), (4, 2, 1)):
    for rep in range(3):
        abi._chk(abi.measure_lib().kt_debug_stream_rows(ctx.h, buf.ptr, N, N, elem, halves, rmw))
    ctx.sync()
    print("rows: elem %d halves %d rmw %d: %d bytes read per launch" % (elem, halves, rmw, N * N * N * elem * halves // 2))

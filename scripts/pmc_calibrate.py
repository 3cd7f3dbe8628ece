"""Launches the calibration streams (known byte counts at tsdf23's access widths); run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kintinuous_amd import abi
ctx = abi.Ctx(0)
GB = 1 << 30
buf = abi.DevBuf(ctx, 2 * GB)
abi._chk(abi.lib().kt_memset(ctx.h, buf.ptr, 1, 2 * GB))
ctx.sync()
for elem, rmw in ((2, 0), (4, 0), (2, 1), (4, 1)):
    for rep in range(3):
        abi._chk(abi.lib().kt_debug_stream(ctx.h, buf.ptr, 2 * GB, elem, rmw))
    ctx.sync()
print("streamed 2 GiB per launch: u16 read, u32 read, u16 rmw, u32 rmw (3 launches each)")

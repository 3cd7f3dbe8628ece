"""kintinuous_amd -- MI355X-native (gfx950) per-frame tracking + fusion hot path of Kintinuous.

The product is the C-ABI shared library libkt_hip.so (include/kt_abi.h, sources under csrc/): hand-written
HIP kernels for the ICP / RGB-D odometry reductions and the cyclical TSDF volume, plus the device-resident
frame pipeline.  This Python package only holds the ctypes binding used by tests and bench.py (abi.py),
the synthetic RGB-D generator (synth.py), .klg I/O (klg.py) and the build helper (build.py).
"""
__all__ = ["abi", "synth", "klg", "build"]

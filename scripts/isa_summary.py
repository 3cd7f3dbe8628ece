"""Per-kernel resource table of libkt_hip.so's device code: compiles every .hip to gfx950 assembly (-S, device only) with the build's
flags and reads the .amdhsa metadata.   python scripts/isa_summary.py > profiles/r03_isa_summary.md   (no GPU needed)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kintinuous_amd import build  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", o) for o in out]


def main():
    tmp = tempfile.mkdtemp()
    rows = []
    for src in build.SOURCES:
        asm = os.path.join(tmp, src + ".s")
        subprocess.check_call([build.hipcc()] + build.FLAGS + ["-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", asm,
                                                           os.path.join(build.CSRC, src)], stderr=subprocess.DEVNULL)
        text = open(asm).read()
        meta = text[text.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in text else ""
        for blk in meta.split("  - .agpr_count:")[1:]:
            get = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
            name = get("name")
            body = re.search(r"^" + re.escape(name) + r":.*?^\.Lfunc_end\d+:", text, re.S | re.M)
            n_valu = len(re.findall(r"^\s+v_", body.group(0), re.M)) if body else 0
            n_salu = len(re.findall(r"^\s+s_", body.group(0), re.M)) if body else 0
            n_mem = len(re.findall(r"^\s+(global_|buffer_|flat_|scratch_)", body.group(0), re.M)) if body else 0
            rows.append((src, name, get("vgpr_count"), get("sgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"),
                         get("group_segment_fixed_size"), get("private_segment_fixed_size"), get("max_flat_workgroup_size"), n_valu, n_salu, n_mem))
    names = demangle([r[1] for r in rows])
    print("# Device-code resources per kernel (gfx950, hipcc " + " ".join(build.FLAGS) + ")\n")
    print("static instruction counts are whole-kernel (all paths), not per iteration\n")
    print("| file | kernel | VGPR | SGPR | VGPR spills | SGPR spills | LDS bytes | scratch bytes | max workgroup | VALU | SALU | VMEM |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for r, n in zip(rows, names):
        print("| " + " | ".join([r[0], "`" + n + "`"] + [str(x) for x in r[2:]]) + " |")


if __name__ == "__main__":
    main()

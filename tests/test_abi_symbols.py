"""The C-ABI library loads without a GPU and exports every symbol include/kt_abi.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _names(path):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kt_[a-z0-9_]+)\s*\(", src)))


def _declared():
    """The boundary: include/kt_abi.h."""
    return _names(os.path.join(ROOT, "include", "kt_abi.h"))


def _declared_debug():
    """Test / analysis hooks, kept OUT of the boundary header: kintinuous_amd/csrc/kt_debug.h."""
    return _names(os.path.join(ROOT, "kintinuous_amd", "csrc", "kt_debug.h"))


def _declared_measure():
    """Measurement kernels, a library of their own (libkt_debug.so): kintinuous_amd/csrc/kt_measure.h."""
    return _names(os.path.join(ROOT, "kintinuous_amd", "csrc", "kt_measure.h"))


def test_header_symbols_are_exported():
    from kintinuous_amd import abi, build
    build.build()
    lib = ctypes.CDLL(abi.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    declared = _declared()
    assert len(declared) > 55
    missing = [n for n in declared + _declared_debug() if not hasattr(lib, n)]
    assert not missing, missing
    assert not [n for n in declared if "debug" in n]          # the product header declares no test hook
    # the measurement kernels are NOT in the product library; libkt_debug.so exports them and loads without a GPU
    assert not [n for n in _declared_measure() if hasattr(lib, n)]
    mlib = ctypes.CDLL(abi.MEASURE_LIB_PATH)
    assert _declared_measure() and not [n for n in _declared_measure() if not hasattr(mlib, n)]
    assert set(abi.MEASURE_SYMBOLS) == set(_declared_measure())


def test_binding_matches_header():
    from kintinuous_amd import abi
    declared = set(_declared()) | set(_declared_debug())
    bound = set(abi.ABI_SYMBOLS)
    assert bound <= declared, sorted(bound - declared)
    assert declared - bound == set(), sorted(declared - bound)


def test_struct_layouts():
    from kintinuous_amd import abi
    assert ctypes.sizeof(abi.Intr) == 16 and ctypes.sizeof(abi.Mat33) == 36
    assert abi.DATATERM_DTYPE.itemsize == 16 and abi.POINT_DTYPE.itemsize == 32
    assert ctypes.sizeof(abi.TrackerConfig) == 18 * 4


def test_no_oracle_in_product_path():
    """The product (package + include) must not reach into oracle/."""
    pkg = os.path.join(ROOT, "kintinuous_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r'#\s*include\s*["<][^">]*oracle', txt), f
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), f
                assert "libkt_oracle" not in txt, f

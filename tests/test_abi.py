"""The C-ABI library loads on a GPU-less host and exports every symbol include/egorear_hip.h declares
(no compute calls here)."""
import ctypes
import os
import re

import pytest

from conftest import REPO


def _declared(header="egorear_hip.h"):
    text = open(os.path.join(REPO, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(egr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    from egorear_amd import hip
    names = _declared()
    assert len(names) >= 15
    lib = ctypes.CDLL(hip.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(hip.EXPORTS) == names


def test_training_header_symbols_are_exported():
    from egorear_amd import hip, hip_train
    names = _declared("egorear_train.h")
    assert len(names) >= 20
    lib = ctypes.CDLL(hip.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(hip_train.TRAIN_EXPORTS) == names


def test_struct_layout_matches_header():
    from egorear_amd import hip
    assert ctypes.sizeof(hip.NMap) == 24
    # 14 leading int32 + 3 nmaps (8-byte aligned) + 5 int32 (+ 4 pad) + 7 int64 group strides + transposed + w_format
    assert ctypes.sizeof(hip.ConvDesc) == 56 + 3 * 24 + 24 + 56 + 8
    assert ctypes.sizeof(hip.W6Job) == 40          # egr_w6_job: two pointers, four int32, one int64
    assert hip.version().startswith("egorear_hip")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "egorear_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, os.path.join(root, f)


def test_headers_are_plain_c():
    """The boundary is a C ABI: both headers must parse as C99 and as C++ on their own (no torch / HIP types in a signature)."""
    import shutil
    import subprocess
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    for cc, lang, std in (("gcc", "c", "-std=c99"), ("g++", "c++", "-std=c++17")):
        if shutil.which(cc) is None:
            pytest.skip(f"{cc} not available")
        for h in ("egorear_hip.h", "egorear_train.h"):
            r = subprocess.run([cc, std, "-fsyntax-only", "-x", lang, "-I" + inc, os.path.join(inc, h)], capture_output=True, text=True)
            assert r.returncode == 0, (cc, h, r.stderr[-2000:])

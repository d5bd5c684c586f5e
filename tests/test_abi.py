"""CPU tests of the C-ABI library: it loads, exports every symbol include/mbk.h declares, its
host-only entry points work, and -- without a GPU -- the compute path refuses loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "mbk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mbk_[a-z_0-9]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from distributedmandelbrot_amd import build, _lib
    so = build.build()
    assert os.path.exists(so)
    lib = C.CDLL(so)
    declared = _declared_symbols()
    assert len(declared) >= 12
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in mbk.h but not exported"
    # the ctypes table binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == declared
    assert _lib.load().mbk_abi_version() == _lib.MBK_ABI_VERSION


def test_struct_layouts_match_header():
    from distributedmandelbrot_amd import _lib
    assert C.sizeof(_lib.mbk_view) == 4 * 8 + 6 * 4
    assert C.sizeof(_lib.mbk_stats) == 4 + 4 + 8 + 8 + 4 + 4 + 8
    assert C.sizeof(_lib.mbk_device_info) == 128 + 64 + 4 * 3 + 4 + 8


def test_geometry_entry_point_is_bit_exact(oracle):
    from distributedmandelbrot_amd.device import MbkError, datachunk_geometry
    for level, ir, ii in [(1, 0, 0), (3, 1, 2), (4, 1, 2), (10, 3, 5), (20, 19, 0), (800000, 251270, 426364)]:
        assert datachunk_geometry(level, ir, ii) == oracle.geometry(level, ir, ii)
    for bad in [(0, 0, 0), (4, 4, 0), (4, 0, 7)]:  # DataChunk.cs:99-106
        with pytest.raises(MbkError):
            datachunk_geometry(*bad)


def test_no_cpu_fallback_without_gpu():
    import distributedmandelbrot_amd as m
    if m.device_count() > 0:
        pytest.skip("a GPU is visible; the refusal path is for GPU-less hosts")
    with pytest.raises(m.MbkError) as e:
        m.MandelbrotDevice(0)
    assert e.value.status == 2 and "no CPU fallback" in str(e.value)
    from distributedmandelbrot_amd import worker
    with pytest.raises(m.MbkError):
        worker.process_workload(4, 256, 0, 0)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "distributedmandelbrot_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "libmandel_oracle" not in text, f

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


def test_header_is_plain_c_and_the_library_links_from_c(tmp_path):
    """include/mbk.h must be consumable by a C compiler (the boundary is a C ABI, not C++), and a C
    program must link against libmbk_hip.so and call the host-only entry points."""
    import shutil
    import subprocess
    from distributedmandelbrot_amd import build
    so = build.build()
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = tmp_path / "abi_smoke.c"
    src.write_text(r"""
#include <stdio.h>
#include "mbk.h"
int main(void) {
    double sr, si, range;
    mbk_view v = {-2.0, -1.5, 3.0, 3.0, 4096u, 4096u, 0u, 0u, 4096u, 4096u};
    mbk_stats st; mbk_device_info info; (void)v; (void)st; (void)info;
    if (mbk_abi_version() != MBK_ABI_VERSION) return 2;
    if (mbk_datachunk_geometry(4u, 1u, 2u, &sr, &si, &range) != MBK_OK) return 3;
    if (mbk_datachunk_geometry(4u, 4u, 0u, &sr, &si, &range) != MBK_ERR_INVALID) return 4;
    mbk_datachunk_geometry(4u, 1u, 2u, &sr, &si, &range);
    int n = -1; int rc = mbk_device_count(&n);
    printf("%.17g %.17g %.17g %d %d %u\n", sr, si, range, rc == MBK_OK || rc == MBK_ERR_NO_DEVICE, n >= 0,
           (unsigned)MBK_CHUNK_BYTES);
    return 0;
}
""")
    exe = tmp_path / "abi_smoke"
    libdir = os.path.dirname(so)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", libdir, "-l:libmbk_hip.so",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["-1", "0", "1", "1", "1", "16777216"]


def test_literal_doubling_guard_matches_a_scan_of_every_row():
    """validate_view's guard for the fma(2, zr*zi, ci) rewrite: a window needs the literal (2*zr)*zi kernels when
    one of its rows has 0 < |ci| < 2^-900 (binary32: 2^-100 after the cast).  Round 5 replaced the scan of every row by two
    binary searches over the monotonic axis; here against the scan (np.linspace IS the axis: tests/test_oracle.py), on axes
    that cross zero with tiny residues, land on zero exactly, stay away from it, run backwards, and on windows that include
    or exclude the crossing and the pinned last sample."""
    import ctypes as C
    import numpy as np
    from distributedmandelbrot_amd import _lib as L
    lib = L.load()
    rs = np.random.RandomState(3)

    def guard(start_i, range_i, h, row0, nrows, f32):
        cv = L.mbk_view(-1.0, start_i, 2.0, range_i, 8, h, 0, row0, 8, nrows)
        out = C.c_int(-1)
        assert lib.mbk_view_needs_literal_doubling(C.byref(cv), L.MBK_PRECISION_F32 if f32 else 0, C.byref(out)) == L.MBK_OK
        return bool(out.value)

    def scan(start_i, range_i, h, row0, nrows, f32):
        ys = np.linspace(start_i, start_i + range_i, h)[row0:row0 + nrows]
        if f32:
            ys = ys.astype(np.float32).astype(np.float64)
        a = np.abs(ys)
        return bool(((a != 0.0) & (a < (2.0 ** -100 if f32 else 2.0 ** -900))).any())

    hits = 0
    for trial in range(3000):
        h = int(rs.choice([1, 2, 3, 7, 64, 513, 4096]))
        kind = trial % 6
        if kind == 0:      # tiny multiples of a tiny unit around zero
            u = 2.0 ** float(rs.randint(-1060, -880))
            start_i, range_i = -u * rs.randint(0, 9), u * rs.randint(1, 40)
        elif kind == 1:    # ordinary axis crossing zero (residues ~1e-17: not tiny)
            start_i, range_i = -rs.uniform(0.1, 2.0), rs.uniform(0.1, 4.0)
        elif kind == 2:    # lands on zero exactly
            n = max(h - 1, 1)
            step = 2.0 ** float(rs.randint(-8, 2))
            start_i, range_i = -step * rs.randint(0, n + 1), step * n
        elif kind == 3:    # backwards, tiny
            u = 2.0 ** float(rs.randint(-1000, -890))
            start_i, range_i = u * rs.randint(0, 9), -u * rs.randint(1, 40)
        elif kind == 4:    # binary32-tiny only
            u = 2.0 ** float(rs.randint(-140, -90))
            start_i, range_i = -u * rs.randint(0, 9), u * rs.randint(1, 40)
        else:              # away from zero
            start_i, range_i = rs.uniform(0.5, 1.0) * rs.choice([-1, 1]) * 2.0, rs.uniform(-0.4, 0.4)
        row0 = int(rs.randint(0, h))
        nrows = int(rs.randint(1, h - row0 + 1))
        for f32 in (False, True):
            want = scan(start_i, range_i, h, row0, nrows, f32)
            assert guard(start_i, range_i, h, row0, nrows, f32) == want, (start_i, range_i, h, row0, nrows, f32)
            hits += want
    assert 300 < hits < 5000, hits

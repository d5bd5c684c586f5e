"""ctypes loader for oracle/libmandel_oracle.so plus an independent pure-numpy restatement.

TEST INFRASTRUCTURE ONLY (see mandel_oracle.c header).  Two restatements of the reference's hot
path live here so that they can be checked against each other before either judges the GPU:

* the C one (``mandel_oracle.c``, compiled ``-ffp-contract=off``), reached through ``COracle``;
* ``numpy_*`` functions below: numpy never contracts a*b+c, and ``np.linspace`` *is* the
  reference's coordinate generator (WorkerCUDA.py:24-32).

Reference citations are relative to /root/reference/, ``WorkerCUDA.py`` =
DistributedMandelbrotWorkerCUDA/DistributedMandelbrotWorkerCUDA.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmandel_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (idempotent).  Returns the .so path."""
    src = os.path.join(_HERE, "mandel_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libmandel_oracle.so"])
    return _SO


class COracle:
    """Thin ctypes view of libmandel_oracle.so."""

    def __init__(self) -> None:
        self.lib = C.CDLL(build())
        L = self.lib
        L.mbo_geometry.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.mbo_geometry.restype = None
        L.mbo_axis.argtypes = [C.c_double, C.c_double, C.c_uint32, C.c_void_p]
        L.mbo_axis.restype = None
        L.mbo_escape.argtypes = [C.c_double, C.c_double, C.c_int32]
        L.mbo_escape.restype = C.c_int32
        L.mbo_quantise.argtypes = [C.c_int32, C.c_uint32]
        L.mbo_quantise.restype = C.c_uint8
        L.mbo_view.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32,
                               C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32,
                               C.c_void_p, C.c_void_p, C.c_int]
        L.mbo_view.restype = C.c_uint64
        L.mbo_view_f32.argtypes = L.mbo_view.argtypes
        L.mbo_view_f32.restype = C.c_uint64
        L.mbo_escape_f32.argtypes = [C.c_float, C.c_float, C.c_int32]
        L.mbo_escape_f32.restype = C.c_int32
        L.mbo_view_smooth.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32,
                                      C.c_int32, C.c_void_p, C.c_void_p]
        L.mbo_view_smooth.restype = None
        L.mbo_datachunk.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_void_p, C.c_void_p, C.c_int]
        L.mbo_datachunk.restype = C.c_uint64
        L.mbo_view_contracted.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32,
                                          C.c_int32, C.c_void_p, C.c_int]
        L.mbo_view_contracted.restype = C.c_uint64
        L.mbo_view_cycle.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int]
        L.mbo_view_cycle.restype = None
        L.mbo_view_cycle_w.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int]
        L.mbo_view_cycle_w.restype = None
        L.mbo_max_threads.restype = C.c_int
        L.mbo_have_avx512.restype = C.c_int
        L.mbo_view_avx512.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_uint32,
                                      C.c_int32, C.c_void_p, C.c_int]
        L.mbo_view_avx512.restype = C.c_uint64

    def max_threads(self) -> int:
        return int(self.lib.mbo_max_threads())

    def geometry(self, level: int, index_real: int, index_imag: int):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self.lib.mbo_geometry(level, index_real, index_imag, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def axis(self, start: float, rng: float, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.float64)
        self.lib.mbo_axis(start, rng, n, out.ctypes.data)
        return out

    def escape(self, cr: float, ci: float, mrd: int) -> int:
        return int(self.lib.mbo_escape(cr, ci, mrd))

    def escape_f32(self, cr: float, ci: float, mrd: int) -> int:
        """The binary32 variant of one pixel (BASELINE cfg4; mbo_escape_f32)."""
        return int(self.lib.mbo_escape_f32(cr, ci, mrd))

    def quantise(self, count: int, mrd: int) -> int:
        return int(self.lib.mbo_quantise(count, mrd))

    def view(self, start_r, start_i, range_r, range_i, width, height, mrd, *, window=None,
             want_counts=True, want_bytes=True, nthreads=0, precision="f64"):
        """Returns (counts int32[nrows,ncols] | None, bytes uint8[nrows,ncols] | None, pixel_iters)."""
        col0, row0, ncols, nrows = window if window is not None else (0, 0, width, height)
        counts = np.empty((nrows, ncols), dtype=np.int32) if want_counts else None
        byts = np.empty((nrows, ncols), dtype=np.uint8) if want_bytes else None
        fn = self.lib.mbo_view_f32 if precision == "f32" else self.lib.mbo_view
        total = fn(start_r, start_i, range_r, range_i, width, height,
                                  col0, row0, ncols, nrows, mrd,
                                  counts.ctypes.data if want_counts else None,
                                  byts.ctypes.data if want_bytes else None, nthreads)
        return counts, byts, int(total)

    def view_smooth(self, start_r, start_i, range_r, range_i, width, height, mrd):
        """(smooth float64[h,w], counts int32[h,w]) -- BASELINE cfg5, see mbo_escape_smooth."""
        smooth = np.empty((height, width), np.float64)
        counts = np.empty((height, width), np.int32)
        self.lib.mbo_view_smooth(start_r, start_i, range_r, range_i, width, height, mrd,
                                 smooth.ctypes.data, counts.ctypes.data)
        return smooth, counts

    def view_contracted(self, start_r, start_i, range_r, range_i, width, height, mrd, *, nthreads=0):
        """Counts under default CUDA-style FMA contraction (see mbo_escape_contracted): a what-if, not the oracle."""
        counts = np.empty((height, width), np.int32)
        self.lib.mbo_view_contracted(start_r, start_i, range_r, range_i, width, height, mrd, counts.ctypes.data, nthreads)
        return counts

    def view_cycle(self, start_r, start_i, range_r, range_i, width, height, mrd, *, first=8, check=16, window_cap=0, nthreads=0):
        """Model of the GPU kernels' cycle test (see mbo_escape_cycle): (counts, executed steps per pixel).  A what-if
        for tests of the claim and for scripts/cycle_model.py, not the oracle.  window_cap: the schedule of
        MBK_OPT_CYCLE_WINDOW (0 = the window of the saved state always doubles, rounds 2-4; the library's default is 32)."""
        counts = np.empty((height, width), np.int32)
        executed = np.empty((height, width), np.int32)
        self.lib.mbo_view_cycle_w(start_r, start_i, range_r, range_i, width, height, mrd, first, check, window_cap,
                                  counts.ctypes.data, executed.ctypes.data, nthreads)
        return counts, executed

    def have_avx512(self) -> bool:
        return bool(self.lib.mbo_have_avx512())

    def view_avx512(self, start_r, start_i, range_r, range_i, width, height, mrd, *, want_counts=True, nthreads=0):
        """The 8-lane AVX-512 evaluation (bench.py's best-effort CPU baseline): (counts | None, pixel_iters)."""
        counts = np.empty((height, width), np.int32) if want_counts else None
        total = self.lib.mbo_view_avx512(start_r, start_i, range_r, range_i, width, height, mrd,
                                         counts.ctypes.data if want_counts else None, nthreads)
        return counts, int(total)

    def datachunk(self, level, mrd, index_real, index_imag, *, want_counts=True, nthreads=0):
        counts = np.empty((4096, 4096), dtype=np.int32) if want_counts else None
        byts = np.empty((4096, 4096), dtype=np.uint8)
        total = self.lib.mbo_datachunk(level, mrd, index_real, index_imag,
                                       counts.ctypes.data if want_counts else None,
                                       byts.ctypes.data, nthreads)
        return counts, byts, int(total)


# --------------------------------------------------------------------------------------------
# Independent numpy restatement (vectorised over pixels; strict IEEE because numpy never fuses).
# --------------------------------------------------------------------------------------------

def numpy_geometry(level: int, index_real: int, index_imag: int):
    """WorkerCUDA.py:75-78 in Python-float arithmetic (what the reference itself executes)."""
    chunk_range = (2 - (-2)) / level
    return -2 + (chunk_range * index_real), -2 + (chunk_range * index_imag), chunk_range


def numpy_axis(start: float, rng: float, n: int) -> np.ndarray:
    """WorkerCUDA.py:24-32: literally np.linspace with the endpoint."""
    return np.linspace(start=start, stop=start + rng, num=n)


def numpy_escape(cr: np.ndarray, ci: np.ndarray, mrd: int, dtype=np.float64) -> np.ndarray:
    """WorkerCUDA.py:39-68, vectorised with an 'alive' mask.  Arrays broadcast.  dtype=np.float32 gives
    the strict-binary32 variant (inputs are rounded to float32 first; numpy keeps float32 arithmetic)."""
    cr, ci = np.broadcast_arrays(np.asarray(cr, np.float64).astype(dtype), np.asarray(ci, np.float64).astype(dtype))
    cr = np.ascontiguousarray(cr).ravel()
    ci = np.ascontiguousarray(ci).ravel()
    out = np.zeros(cr.shape, np.int32)
    idx = np.arange(cr.size)
    zr, zi, kr, ki = cr.copy(), ci.copy(), cr.copy(), ci.copy()
    with np.errstate(all="ignore"):
        for n in range(1, mrd):
            if idx.size == 0:
                break
            t = zr * zr - zi * zi
            u = (dtype(2) * zr) * zi
            zr = t + kr
            zi = u + ki
            m = zr * zr + zi * zi
            esc = m >= dtype(4)
            if esc.any():
                out[idx[esc]] = n
                keep = ~esc
                idx, zr, zi, kr, ki = idx[keep], zr[keep], zi[keep], kr[keep], ki[keep]
    return out


def numpy_quantise(counts: np.ndarray, mrd: int) -> np.ndarray:
    """WorkerCUDA.py:96-98 verbatim arithmetic."""
    with np.errstate(all="ignore"):
        out = (np.asarray(counts).astype(np.float64) * 256) / mrd
        return np.ceil(out).astype(np.uint8)


def numpy_view(start_r, start_i, range_r, range_i, width, height, mrd, window=None):
    col0, row0, ncols, nrows = window if window is not None else (0, 0, width, height)
    xr = numpy_axis(start_r, range_r, width)[col0:col0 + ncols]
    xi = numpy_axis(start_i, range_i, height)[row0:row0 + nrows]
    counts = numpy_escape(xr[None, :], xi[:, None], mrd).reshape(nrows, ncols)
    return counts, numpy_quantise(counts, mrd)


def pixel_iterations(counts: np.ndarray, mrd: int) -> int:
    """iters(p) = count if count > 0 else mrd - 1."""
    c = np.asarray(counts).astype(np.int64)
    return int(np.where(c > 0, c, max(mrd - 1, 0)).sum())

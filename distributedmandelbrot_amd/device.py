"""MandelbrotDevice -- one GPU context over the C ABI (include/mbk.h).

Mirrors, for a generic view, what the reference's ``process_workload`` does for a DataChunk tile
(DistributedMandelbrotWorkerCUDA.py:70-100, "WorkerCUDA.py"): coordinates (gen_arrays, :19-37),
the escape-time kernel (calc_mb_value, :39-68) and the uint8 quantiser (:96-98) -- all on the GPU.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import _lib as L


class MbkError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"libmbk_hip status {status}: {message}")
        self.status = status


@dataclass(frozen=True)
class View:
    """width x height samples of [start_r, start_r+range_r] x [start_i, start_i+range_i], endpoints
    included (np.linspace semantics, WorkerCUDA.py:24-32)."""
    start_r: float
    start_i: float
    range_r: float
    range_i: float
    width: int
    height: int

    @staticmethod
    def centered(center_r: float, center_i: float, span: float, width: int, height: Optional[int] = None):
        height = width if height is None else height
        return View(center_r - span / 2, center_i - span / 2, span, span, width, height)


@dataclass
class TileStats:
    kernel_ms: float
    d2h_ms: float
    pixel_iterations: int
    never_pixels: int
    all_bytes_zero: bool   # DataChunk.IsNeverChunk (DataChunk.cs:82)
    all_bytes_one: bool    # DataChunk.IsImmediateChunk (DataChunk.cs:87)
    rle_runs: int = 0      # runs of equal bytes: RLE codec size = 1 + 5*rle_runs (DataChunkSerializer.cs:56-100)


def device_count() -> int:
    lib = L.load()
    n = C.c_int(0)
    st = lib.mbk_device_count(C.byref(n))
    if st != L.MBK_OK:
        return 0
    return n.value


def datachunk_geometry(level: int, index_real: int, index_imag: int) -> Tuple[float, float, float]:
    """(start_r, start_i, range) of a DataChunk tile: WorkerCUDA.py:75-78 == DataChunk.cs:32-33,59-66."""
    lib = L.load()
    a, b, c = C.c_double(), C.c_double(), C.c_double()
    st = lib.mbk_datachunk_geometry(level, index_real, index_imag, C.byref(a), C.byref(b), C.byref(c))
    if st != L.MBK_OK:
        raise MbkError(st, (lib.mbk_last_error(None) or b"").decode())
    return a.value, b.value, c.value


class MandelbrotDevice:
    """One mbk_ctx == one GPU.  Not thread-safe: use one host thread per instance."""

    SLOTS = L.MBK_SLOTS   # tiles in flight of the host-buffer API (submit_* / wait)
    WORKER_DEPTH = L.MBK_WORKER_DEPTH   # what the worker loops keep in flight (include/mbk.h)

    def __init__(self, device: int = 0):
        self._lib = L.load()
        h = C.c_void_p()
        st = self._lib.mbk_create(device, C.byref(h))
        if st != L.MBK_OK:
            raise MbkError(st, (self._lib.mbk_last_error(None) or b"").decode())
        self._h = h
        self.device = device
        self._pinned = []
        self._ser_buf = np.empty(1 + L.MBK_CHUNK_BYTES, np.uint8)   # reused by serialize_last

    # -- lifecycle -------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            for p in self._pinned:
                self._lib.mbk_host_free(self._h, p)
            self._pinned = []
            self._lib.mbk_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int) -> None:
        if st != L.MBK_OK:
            raise MbkError(st, (self._lib.mbk_last_error(self._h) or b"").decode())

    # -- queries ---------------------------------------------------------------------------
    def info(self) -> dict:
        inf = L.mbk_device_info()
        self._check(self._lib.mbk_get_device_info(self._h, C.byref(inf)))
        return {"name": inf.name.decode(), "arch": inf.arch.decode(),
                "compute_units": inf.compute_units, "clock_mhz": inf.clock_mhz,
                "wavefront_size": inf.wavefront_size, "total_mem": inf.total_mem}

    def pci_bus_id(self) -> str:
        buf = C.create_string_buffer(32)
        self._check(self._lib.mbk_device_pci_bus_id(self._h, buf, 32))
        return buf.value.decode()

    def pinned_empty(self, shape, dtype) -> np.ndarray:
        """A numpy array over pinned host memory (freed when the device is closed)."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._check(self._lib.mbk_host_alloc(self._h, max(n, 1), C.byref(p)))
        self._pinned.append(p)
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def set_option(self, name: str, value: int) -> None:
        """Tuning option (include/mbk.h enum mbk_option; names in _lib.OPTIONS).  Scheduling only:
        every accepted value gives bit-identical results."""
        self._check(self._lib.mbk_set_option(self._h, L.OPTIONS[name], int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_uint32(0)
        self._check(self._lib.mbk_get_option(self._h, L.OPTIONS[name], C.byref(v)))
        return int(v.value)

    def scan_occupancy(self) -> dict:
        """hipOccupancyMaxActiveBlocksPerMultiprocessor of the scan-path kernels (single-wave workgroups per CU)."""
        out = {}
        for k, name in enumerate(["f64_scan", "f64_heavy", "f32_scan", "f32_heavy"]):
            v = C.c_uint32(0)
            self._check(self._lib.mbk_get_option(self._h, L.MBK_INFO_SCAN_WG_PER_CU + k, C.byref(v)))
            out[name] = int(v.value)
        return out

    def xcd_shares(self) -> dict:
        """MBK_OPT_XCD_BALANCE = 1: the shares of the heavy list the eight XCDs currently get (an even deal is 0.125 each), the
        number of the last units launch whose time stamps were read, and the units launches issued."""
        vals = []
        for k in range(10):
            v = C.c_uint32(0)
            self._check(self._lib.mbk_get_option(self._h, L.MBK_INFO_XCD_SHARE + k, C.byref(v)))
            vals.append(int(v.value))
        return {"shares": [round(v / 1048576.0, 5) for v in vals[:8]], "last_launch_read": vals[8], "units_launches": vals[9]}

    def spill_info(self) -> dict:
        """SPILL (MBK_OPT_SPILL_FIRST): lanes the last launch with a second pass handed over to it, and how many launches of
        this context ran with one."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.mbk_get_option(self._h, L.MBK_INFO_SPILL, C.byref(a)))
        self._check(self._lib.mbk_get_option(self._h, L.MBK_INFO_SPILL + 1, C.byref(b)))
        return {"lanes_last_launch": int(a.value), "launches": int(b.value)}

    def quantise_counts(self, counts: np.ndarray, mrd: int) -> np.ndarray:
        """The device's quantiser alone (WorkerCUDA.py:96-98) on host int32 counts in [0, mrd-1]."""
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        out = np.empty(counts.shape, np.uint8)
        self._check(self._lib.mbk_quantise_counts(self._h, counts.ctypes.data, counts.size, mrd, out.ctypes.data))
        return out

    # -- compute ---------------------------------------------------------------------------
    @staticmethod
    def _cview(view: View, window) -> L.mbk_view:
        col0, row0, ncols, nrows = window if window is not None else (0, 0, view.width, view.height)
        return L.mbk_view(view.start_r, view.start_i, view.range_r, view.range_i,
                          view.width, view.height, col0, row0, ncols, nrows)

    def compute_view(self, view: View, mrd: int, *, window=None, want_counts: bool = True,
                     want_bytes: bool = True, kernel: str = "default", precision: str = "f64",
                     out_counts: Optional[np.ndarray] = None, out_bytes: Optional[np.ndarray] = None):
        """Synchronous: returns (counts int32[nrows,ncols] | None, bytes uint8[nrows,ncols] | None, TileStats)."""
        cv = self._cview(view, window)
        shape = (cv.nrows, cv.ncols)
        flags = L.KERNELS[kernel] | L.PRECISIONS[precision]
        counts = byts = None
        if want_counts:
            counts = out_counts if out_counts is not None else np.empty(shape, np.int32)
            assert counts.dtype == np.int32 and counts.size == shape[0] * shape[1] and counts.flags.c_contiguous
            flags |= L.MBK_WANT_COUNTS
        if want_bytes:
            byts = out_bytes if out_bytes is not None else np.empty(shape, np.uint8)
            assert byts.dtype == np.uint8 and byts.size == shape[0] * shape[1] and byts.flags.c_contiguous
            flags |= L.MBK_WANT_BYTES
        st = L.mbk_stats()
        self._check(self._lib.mbk_view_compute(
            self._h, C.byref(cv), mrd, flags,
            counts.ctypes.data if counts is not None else None,
            byts.ctypes.data if byts is not None else None, C.byref(st)))
        return counts, byts, _stats(st)

    def datachunk(self, level: int, mrd: int, index_real: int, index_imag: int, *,
                  want_counts: bool = False, out_bytes: Optional[np.ndarray] = None):
        """The reference's process_workload (WorkerCUDA.py:70-100): uint8[16777216] for one tile.
        Returns (bytes uint8[16777216], counts int32[16777216] | None, TileStats)."""
        byts = out_bytes if out_bytes is not None else np.empty(L.MBK_CHUNK_BYTES, np.uint8)
        assert byts.dtype == np.uint8 and byts.size == L.MBK_CHUNK_BYTES and byts.flags.c_contiguous
        counts = np.empty(L.MBK_CHUNK_BYTES, np.int32) if want_counts else None
        st = L.mbk_stats()
        self._check(self._lib.mbk_datachunk(
            self._h, level, mrd, index_real, index_imag, byts.ctypes.data,
            counts.ctypes.data if counts is not None else None, C.byref(st)))
        return byts, counts, _stats(st)

    def compute_view_smooth(self, view: View, mrd: int, *, window=None, kernel: str = "default"):
        """BASELINE cfg5 (not in the reference): continuous escape-time value nu = n + 1 - log2(0.5 ln|z_n|^2)
        at the reference's bailout, 0 for never-escaped pixels.
        Returns (smooth float64[nrows,ncols], counts int32[nrows,ncols], TileStats)."""
        cv = self._cview(view, window)
        shape = (cv.nrows, cv.ncols)
        smooth = np.empty(shape, np.float64)
        counts = np.empty(shape, np.int32)
        st = L.mbk_stats()
        self._check(self._lib.mbk_view_compute_smooth(self._h, C.byref(cv), mrd, L.KERNELS[kernel],
                                                      counts.ctypes.data, smooth.ctypes.data, C.byref(st)))
        return smooth, counts, _stats(st)

    def launch_view_smooth(self, view: View, mrd: int, *, d_smooth: int, d_counts: int = 0, stream: int = 0,
                           window=None, kernel: str = "default") -> None:
        cv = self._cview(view, window)
        self._check(self._lib.mbk_view_launch_smooth(self._h, C.byref(cv), mrd, L.KERNELS[kernel],
                                                     d_counts or None, d_smooth, stream or None))

    def serialize_last(self) -> Tuple[bytes, int]:
        """The last tile's quantised bytes exactly as DataChunk.Serialize (DataChunk.cs:173-206) would
        write them (code byte + Raw or RLE payload, the shorter; Raw on ties), encoded on the GPU.
        Returns (stream, codec)."""
        cap = len(self._ser_buf)
        size, codec = C.c_uint64(0), C.c_uint32(0)
        st = self._lib.mbk_serialize_last(self._h, self._ser_buf.ctypes.data, cap, C.byref(size), C.byref(codec))
        if st == L.MBK_ERR_INVALID and size.value > cap:
            self._ser_buf = np.empty(size.value, np.uint8)
            st = self._lib.mbk_serialize_last(self._h, self._ser_buf.ctypes.data, size.value, C.byref(size),
                                              C.byref(codec))
        self._check(st)
        return self._ser_buf[:size.value].tobytes(), int(codec.value)

    def submit_datachunk(self, slot: int, level: int, mrd: int, index_real: int, index_imag: int,
                         out_bytes: np.ndarray, lazy_uniform: bool = False) -> None:
        """Enqueue a tile on `slot` (0 .. SLOTS-1) and return at once; `out_bytes` (uint8[16777216], ideally from
        pinned_empty) is valid after wait(slot).  Several slots = the D2H of one tile overlaps the kernels of the others.
        lazy_uniform (MBK_LAZY_UNIFORM): the caller does not need `out_bytes` when the tile turns out all-0 / all-1 -- the
        TileStats returned by wait() say which constant it is, and `out_bytes` is then unspecified: a tile wholly outside
        |c| = 2 costs no GPU work at all, the copy of a tile the host probe takes for all-exterior is decided when its
        statistics arrive, every other tile is copied as usual."""
        assert out_bytes.dtype == np.uint8 and out_bytes.size == L.MBK_CHUNK_BYTES and out_bytes.flags.c_contiguous
        self._check(self._lib.mbk_datachunk_submit_ex(self._h, slot, level, mrd, index_real, index_imag,
                                                      out_bytes.ctypes.data, None,
                                                      L.MBK_LAZY_UNIFORM if lazy_uniform else 0))

    def submit_view(self, slot: int, view: View, mrd: int, *, window=None, out_counts: Optional[np.ndarray] = None,
                    out_bytes: Optional[np.ndarray] = None, kernel: str = "default", precision: str = "f64") -> None:
        """Enqueue a view / window on `slot` and return at once; the given host arrays (C-contiguous, sized
        for the window; slices of a larger image are fine) are valid after wait(slot)."""
        cv = self._cview(view, window)
        n = cv.nrows * cv.ncols
        flags = L.KERNELS[kernel] | L.PRECISIONS[precision]
        for arr, dt, flag in ((out_counts, np.int32, L.MBK_WANT_COUNTS), (out_bytes, np.uint8, L.MBK_WANT_BYTES)):
            if arr is not None:
                assert arr.dtype == dt and arr.size == n and arr.flags.c_contiguous
                flags |= flag
        self._check(self._lib.mbk_view_submit(
            self._h, slot, C.byref(cv), mrd, flags,
            out_counts.ctypes.data if out_counts is not None else None,
            out_bytes.ctypes.data if out_bytes is not None else None))

    def wait(self, slot: int) -> TileStats:
        st = L.mbk_stats()
        self._check(self._lib.mbk_wait(self._h, slot, C.byref(st)))
        return _stats(st)

    def launch_view(self, view: View, mrd: int, *, d_counts: int = 0, d_bytes: int = 0,
                    stream: int = 0, window=None, kernel: str = "default", precision: str = "f64") -> None:
        """Asynchronous launch on raw DEVICE pointers (e.g. torch tensors' data_ptr()) on ``stream``
        (a hipStream_t as int; 0 = HIP's null stream, which is also torch's default stream)."""
        cv = self._cview(view, window)
        flags = (L.KERNELS[kernel] | L.PRECISIONS[precision] | (L.MBK_WANT_COUNTS if d_counts else 0)
                 | (L.MBK_WANT_BYTES if d_bytes else 0))
        self._check(self._lib.mbk_view_launch(self._h, C.byref(cv), mrd, flags,
                                              d_counts or None, d_bytes or None, stream or None))

    def reduce_counts(self, d_counts: int, n: int, mrd: int, stream: int = 0) -> TileStats:
        st = L.mbk_stats()
        self._check(self._lib.mbk_reduce_counts(self._h, d_counts, n, mrd, stream or None, C.byref(st)))
        return _stats(st)


def _stats(st: L.mbk_stats) -> TileStats:
    return TileStats(float(st.kernel_ms), float(st.d2h_ms), int(st.pixel_iterations),
                     int(st.never_pixels), bool(st.all_bytes_zero), bool(st.all_bytes_one), int(st.rle_runs))

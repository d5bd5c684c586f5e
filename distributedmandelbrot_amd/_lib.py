"""ctypes binding of libmbk_hip.so (C ABI: include/mbk.h).

The library is built in-tree by ``distributedmandelbrot_amd.build`` and loaded from this directory.
If it is missing the import of the product path FAILS LOUDLY -- there is no CPU fallback
(the CPU oracle under oracle/ is test infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libmbk_hip.so")

MBK_OK, MBK_ERR_INVALID, MBK_ERR_NO_DEVICE, MBK_ERR_HIP, MBK_ERR_NOMEM, MBK_ERR_NET = range(6)
MBK_WANT_COUNTS = 0x1
MBK_WANT_BYTES = 0x2
MBK_KERNEL_DEFAULT = 0x000
MBK_KERNEL_SIMPLE = 0x100
MBK_KERNEL_ASM = 0x200
MBK_KERNEL_REFILL = 0x300
MBK_KERNEL_GROUP = 0x400
MBK_KERNEL_SCAN = 0x500
KERNELS = {"default": MBK_KERNEL_DEFAULT, "simple": MBK_KERNEL_SIMPLE, "asm": MBK_KERNEL_ASM,
           "refill": MBK_KERNEL_REFILL, "group": MBK_KERNEL_GROUP, "scan": MBK_KERNEL_SCAN}
# enum mbk_option (include/mbk.h), in order
OPTIONS = {name: i for i, name in enumerate(
    ["order", "waves_per_wg", "group_steps", "exact_steps", "probe_steps", "scan_waves", "scan_xcd_map", "scan_col_period", "heavy_share",
     "rf_livemin", "rf_patience", "rf_batch", "rf_waves", "cycle_detect", "probe_mid", "prepass_overlap", "exact_long", "scan_inline", "wave_limit", "units_min_light", "xcd_balance", "m_late", "h_settled", "classify_wg", "scan_strip", "cycle_window", "spill_first", "spill_lanes", "spill_min_mrd", "spill_min_blocks", "spill_cyc_shift"])}
MBK_PRECISION_F32 = 0x1000
MBK_LAZY_UNIFORM = 0x2000
PRECISIONS = {"f64": 0, "f32": MBK_PRECISION_F32}
MBK_SLOTS = 4
MBK_WORKER_DEPTH = 3
MBK_INFO_SCAN_WG_PER_CU = 100
MBK_INFO_XCD_SHARE = 110
MBK_INFO_SPILL = 130
MBK_CODEC_RAW = 0x00
MBK_CODEC_RLE = 0x01
MBK_CHUNK_DEFINITION = 4096
MBK_CHUNK_BYTES = 4096 * 4096
MBK_ABI_VERSION = 4


class mbk_view(C.Structure):
    _fields_ = [("start_r", C.c_double), ("start_i", C.c_double),
                ("range_r", C.c_double), ("range_i", C.c_double),
                ("width", C.c_uint32), ("height", C.c_uint32),
                ("col0", C.c_uint32), ("row0", C.c_uint32),
                ("ncols", C.c_uint32), ("nrows", C.c_uint32)]


class mbk_stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_float), ("d2h_ms", C.c_float),
                ("pixel_iterations", C.c_uint64), ("never_pixels", C.c_uint64),
                ("all_bytes_zero", C.c_uint32), ("all_bytes_one", C.c_uint32),
                ("rle_runs", C.c_uint64)]


class mbk_device_info(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("arch", C.c_char * 64),
                ("compute_units", C.c_int), ("clock_mhz", C.c_int),
                ("wavefront_size", C.c_int), ("total_mem", C.c_uint64)]


class mbk_worker_report(C.Structure):
    _fields_ = [("leased", C.c_uint64), ("accepted", C.c_uint64), ("rejected", C.c_uint64), ("resets", C.c_uint64),
                ("uniform_tiles", C.c_uint64), ("pixel_iterations", C.c_uint64),
                ("kernel_ms_sum", C.c_double), ("seconds", C.c_double), ("net_retries", C.c_uint64)]


# enum mbk_net_option (include/mbk.h), in order: process-wide network behaviour of the native worker loop
NET_OPTIONS = {name: i for i, name in enumerate(
    ["max_connections", "connect_timeout_ms", "io_timeout_ms", "retries", "backoff_ms", "stop", "peak_connections", "feeder_slots"])}


FEEDER_SUBMIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p)
FEEDER_WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(mbk_stats))
FEEDER_ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64)
FEEDER_RELEASE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
FEEDER_ON_TILE = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint32 * 4), C.POINTER(mbk_stats), C.c_int)


class mbk_feeder_ops(C.Structure):
    _fields_ = [("user", C.c_void_p), ("submit", FEEDER_SUBMIT), ("wait", FEEDER_WAIT), ("alloc", FEEDER_ALLOC),
                ("release", FEEDER_RELEASE), ("on_tile", FEEDER_ON_TILE)]


# symbol -> (restype, argtypes); this table is also what tests/test_abi.py checks against mbk.h
SIGNATURES = {
    "mbk_abi_version": (C.c_int, []),
    "mbk_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "mbk_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "mbk_destroy": (None, [C.c_void_p]),
    "mbk_last_error": (C.c_char_p, [C.c_void_p]),
    "mbk_get_device_info": (C.c_int, [C.c_void_p, C.POINTER(mbk_device_info)]),
    "mbk_device_pci_bus_id": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "mbk_host_alloc": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]),
    "mbk_host_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mbk_datachunk_geometry": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.POINTER(C.c_double)]),
    "mbk_view_outside_circle": (C.c_int, [C.POINTER(mbk_view), C.c_uint32, C.POINTER(C.c_int)]),
    "mbk_view_needs_literal_doubling": (C.c_int, [C.POINTER(mbk_view), C.c_uint32, C.POINTER(C.c_int)]),
    "mbk_view_launch": (C.c_int, [C.c_void_p, C.POINTER(mbk_view), C.c_uint32, C.c_uint32,
                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "mbk_view_compute": (C.c_int, [C.c_void_p, C.POINTER(mbk_view), C.c_uint32, C.c_uint32,
                                   C.c_void_p, C.c_void_p, C.POINTER(mbk_stats)]),
    "mbk_datachunk": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                C.c_void_p, C.c_void_p, C.POINTER(mbk_stats)]),
    "mbk_view_launch_smooth": (C.c_int, [C.c_void_p, C.POINTER(mbk_view), C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "mbk_view_compute_smooth": (C.c_int, [C.c_void_p, C.POINTER(mbk_view), C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.POINTER(mbk_stats)]),
    "mbk_datachunk_submit": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                       C.c_void_p, C.c_void_p]),
    "mbk_datachunk_submit_ex": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_void_p, C.c_void_p, C.c_uint32]),
    "mbk_view_submit": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(mbk_view), C.c_uint32, C.c_uint32,
                                  C.c_void_p, C.c_void_p]),
    "mbk_wait": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(mbk_stats)]),
    "mbk_serialize_last": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64),
                                     C.POINTER(C.c_uint32)]),
    "mbk_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32]),
    "mbk_get_option": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint32)]),
    "mbk_quantise_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]),
    "mbk_units_plan": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mbk_units_lookup": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]),
    "mbk_reduce_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p,
                                    C.POINTER(mbk_stats)]),
    "mbk_worker_run": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint16, C.c_uint64, C.c_uint32,
                                 C.POINTER(mbk_worker_report)]),
    "mbk_net_set_option": (C.c_int, [C.c_int, C.c_uint32]),
    "mbk_net_get_option": (C.c_int, [C.c_int, C.POINTER(C.c_uint32)]),
    "mbk_feeder_run": (C.c_int, [C.POINTER(mbk_feeder_ops), C.c_char_p, C.c_uint16, C.c_uint64, C.c_uint32,
                                 C.POINTER(mbk_worker_report)]),
}

_lib = None


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Whichever
    copy is loaded first serves the whole process; torch fails with "No HIP GPUs are available" when
    the system copy got in first.  So: if torch is installed but not imported yet, pre-load ITS copy
    (cheap -- torch itself is not imported) so that either import order works.  Opt out with
    MBK_HIP_RUNTIME=system."""
    if os.environ.get("MBK_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> C.CDLL:
    """Load libmbk_hip.so and declare every entry point.  Raises if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: build it with `python -m distributedmandelbrot_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(SO_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.mbk_abi_version() != MBK_ABI_VERSION:
        raise ImportError("libmbk_hip.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib

#!/usr/bin/env python3
"""Generate tests/golden/reference_vectors.npz by EXECUTING THE REFERENCE'S OWN SOURCE.

Run once, in the build container (needs /root/reference; the GPU box does not have it):

    python tests/golden/make_golden.py

The reference's only implementation of the hot path is
/root/reference/DistributedMandelbrotWorkerCUDA/DistributedMandelbrotWorkerCUDA.py
("WorkerCUDA.py").  It cannot be imported as-is: it needs numba and a CUDA driver (line 3-4, 39).
This script installs a *shim* ``numba`` module whose ``vectorize`` applies the decorated Python
function ``calc_mb_value`` (WorkerCUDA.py:39-68) element by element under CPython, with Python
floats (IEEE-754 binary64, every operation individually rounded -- CPython never contracts), and
whose ``cuda.to_device`` / ``cuda.device_array`` are host no-ops.  Everything else -- ``gen_arrays``
(np.linspace/tile/repeat, :19-37), the tile geometry (:75-78) and the quantiser (:96-98) -- is the
reference's unmodified code running on the numpy installed here (2.2.6).

So the vectors pin "the strict-IEEE reading of the reference's source".  What they cannot pin is
whatever FMA contraction numba/NVVM applied on the author's GPU (numba and its flags are
un-vendored and un-pinned).

Nothing from the reference is copied into this repository: the module is loaded from where it
lies and only its outputs are stored.
"""
from __future__ import annotations

import hashlib
import importlib.util
import multiprocessing as mp
import os
import sys
import types

import numpy as np

REF = "/root/reference/DistributedMandelbrotWorkerCUDA/DistributedMandelbrotWorkerCUDA.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.npz")

_PYFUNC = None          # the undecorated reference scalar function
_LAST_OUT = {}          # last int32 array the "kernel" wrote (so counts can be recorded too)


def _chunk(args):
    r, i, mrd = args
    f = _PYFUNC
    return [f(a, b, mrd) for a, b in zip(r, i)]


def _install_numba_shim():
    numba = types.ModuleType("numba")
    cuda = types.ModuleType("numba.cuda")
    cuda.to_device = lambda a: a
    cuda.device_array = lambda shape, dtype: np.empty(shape, dtype=dtype)

    def vectorize(signatures, target=None):
        assert signatures == ["int32(float64, float64, int32)"] and target == "cuda"

        def deco(pyfunc):
            global _PYFUNC
            _PYFUNC = pyfunc

            def ufunc(r, i, mrd, out=None):
                r = np.asarray(r, dtype=np.float64).ravel().tolist()   # python floats
                i = np.asarray(i, dtype=np.float64).ravel().tolist()
                mrd = int(np.int32(mrd))
                n = len(r)
                if n > 1 << 16:
                    step = 1 << 16
                    jobs = [(r[k:k + step], i[k:k + step], mrd) for k in range(0, n, step)]
                    with mp.get_context("fork").Pool(os.cpu_count()) as pool:
                        parts = pool.map(_chunk, jobs, chunksize=1)
                    res = np.fromiter((v for p in parts for v in p), dtype=np.int32, count=n)
                else:
                    res = np.array(_chunk((r, i, mrd)), dtype=np.int32)
                if out is None:
                    out = np.empty(n, dtype=np.int32)
                out[...] = res.reshape(out.shape)
                _LAST_OUT["counts"] = out
                return out

            return ufunc

        return deco

    numba.vectorize = vectorize
    numba.cuda = cuda
    sys.modules["numba"] = numba
    sys.modules["numba.cuda"] = cuda


def _load_reference():
    _install_numba_shim()
    spec = importlib.util.spec_from_file_location("reference_worker", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    # cuda.device_array returns an array exposing copy_to_host() in numba; give ndarray one.
    class _Arr(np.ndarray):
        def copy_to_host(self):
            return np.asarray(self).copy()
    mod.cuda.device_array = lambda shape, dtype: np.empty(shape, dtype=dtype).view(_Arr)
    return mod


# Small windows evaluated through the reference's gen_arrays + calc_mb_value:
# (name, start_r, start_i, range, definition, mrd)
SMALL = [
    ("cfg1_full_set_64", -2.0, -1.5, 3.0, 64, 256),
    ("cfg1_full_set_96_mrd1000", -2.0, -1.5, 3.0, 96, 1000),
    ("level1_tile_64", -2.0, -2.0, 4.0, 64, 256),
    ("level4_1_2_48", -1.0, 0.0, 1.0, 48, 256),
    ("seahorse_48", -0.755, 0.10, 0.02, 48, 1024),
    ("cfg3_deep_zoom_24", -0.743648, 0.131820, 1e-5, 24, 10000),
    ("needle_real_axis_33", -2.0, -0.01, 0.02, 33, 300),
    ("tiny_mrd2_16", -2.0, -2.0, 4.0, 16, 2),
    ("tiny_mrd1_8", -2.0, -2.0, 4.0, 8, 1),
    ("single_pixel", -0.75, 0.1, 0.5, 1, 100),
]

# Known-answer points: (cr, ci, mrd)
POINTS = [
    (0.0, 0.0, 256), (-2.0, 0.0, 256), (2.0, 2.0, 256), (-0.75, 0.0, 256), (-0.75, 0.0, 10000),
    (0.25, 0.0, 256), (0.25, 0.0, 100000), (0.26, 0.0, 1000), (-1.75, 0.0, 1000), (0.0, 1.0, 1000),
    (-2.0, 1e-9, 1000), (0.3, 0.5, 1000), (-0.1, 0.651, 5000), (-1.999999, 0.0, 5000),
    (1e-300, 1e-300, 50), (-0.5, 0.0, 0), (-0.5, 0.0, 1), (3.0, 0.0, 2),
]

# Full 4096x4096 tiles through the reference's process_workload, unmodified:
# (level, mrd, index_real, index_imag).  Chosen from the levels/mrds of the reference's own
# launch profiles (Properties/launchSettings.json:5,9: -l 4:256,10:1024,20:1024).
FULL = [
    (4, 256, 0, 0),        # all exterior: cheap under CPython
    (10, 1024, 0, 5),      # [-2,-1.6] x [0,0.4]: row 0 is the real-axis antenna (never escapes), slow escapes above it
    (4, 256, 1, 2),        # [-1,0] x [0,1]: cardioid + period-2 bulb + boundary filaments (minutes under CPython)
    (1, 256, 0, 0),        # the whole [-2,2]^2 image (SURVEY.md section 4 KAT list)
    (10, 1024, 3, 5),      # [-0.8,-0.4] x [0,0.4]: bulb boundary, ~half in-set (SURVEY KAT list; ~10 min)
    (20, 1024, 7, 9),      # [-0.6,-0.4] x [-0.2,0]: inside the cardioid, every pixel runs 1023 steps (~15 min)
]
if os.environ.get("GOLDEN_SKIP_EXPENSIVE"):
    FULL = FULL[:2]


def main():
    ref = _load_reference()
    data = {}
    names = []
    for name, sr, si, rng, n, mrd in SMALL:
        r_rep, i_rep = ref.gen_arrays(start_r=sr, start_i=si, _range=rng, definition=n)
        counts = ref.calc_mb_value(r_rep, i_rep, mrd)
        data[f"small/{name}/params"] = np.array([sr, si, rng, n, mrd], dtype=np.float64)
        data[f"small/{name}/counts"] = np.asarray(counts, dtype=np.int32).reshape(n, n)
        # the reference's coordinate arrays themselves (first row / first column) pin linspace
        data[f"small/{name}/axis_r"] = np.asarray(r_rep[:n], dtype=np.float64)
        data[f"small/{name}/axis_i"] = np.asarray(i_rep[::n], dtype=np.float64)
        names.append(name)
        print("small", name, "mean", counts.mean())
    data["small/names"] = np.array(names)

    pts = np.array(POINTS, dtype=np.float64)
    data["points/inputs"] = pts
    data["points/counts"] = np.array([_PYFUNC(float(a), float(b), int(m)) for a, b, m in POINTS],
                                     dtype=np.int32)
    print("points", data["points/counts"])

    full_names = []
    for level, mrd, ir, ii in FULL:
        out = ref.process_workload(level, mrd, ir, ii)          # uint8[16777216]
        counts = np.asarray(_LAST_OUT["counts"]).copy()          # int32[16777216]
        key = f"{level}_{mrd}_{ir}_{ii}"
        data[f"full/{key}/params"] = np.array([level, mrd, ir, ii], dtype=np.int64)
        data[f"full/{key}/bytes_sha256"] = np.array(hashlib.sha256(out.tobytes()).hexdigest())
        data[f"full/{key}/counts_sha256"] = np.array(
            hashlib.sha256(counts.astype("<i4").tobytes()).hexdigest())
        data[f"full/{key}/bytes_sub64"] = out.reshape(4096, 4096)[::64, ::64].copy()
        data[f"full/{key}/counts_sub64"] = counts.reshape(4096, 4096)[::64, ::64].copy()
        data[f"full/{key}/bytes_row2048"] = out.reshape(4096, 4096)[2048].copy()
        data[f"full/{key}/counts_sum"] = np.array(counts.astype(np.int64).sum())
        data[f"full/{key}/zeros"] = np.array(int((counts == 0).sum()))
        full_names.append(key)
        print("full", key, "bytes sha", data[f"full/{key}/bytes_sha256"], "zeros",
              data[f"full/{key}/zeros"])
    data["full/names"] = np.array(full_names)
    data["meta/numpy_version"] = np.array(np.__version__)
    np.savez_compressed(OUT, **data)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()

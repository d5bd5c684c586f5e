"""How many pixels of a tile change if the worker's compiler contracts a*b+c into FMA (what numba/NVVM does by
default on the reference author's GPU, and what nobody can pin: SURVEY.md section 0)?  For the six golden
DataChunk tiles: strict evaluation (the parity target, = this worker) vs the contracted what-if
(oracle/mandel_oracle.c: mbo_escape_contracted).  CPU only.  Writes profiles/r02/contraction_sensitivity.json.
    python scripts/contraction_table.py"""
import json, os, sys, time
sys.path.insert(0, ".")
import numpy as np
from oracle.oracle import COracle, numpy_quantise

o = COracle()
tiles = [(4, 256, 0, 0), (10, 1024, 0, 5), (4, 256, 1, 2), (1, 256, 0, 0), (10, 1024, 3, 5), (20, 1024, 7, 9)]
rows = []
for level, mrd, ir, ii in tiles:
    t0 = time.time()
    sr, si, rng = o.geometry(level, ir, ii)
    strict, _, _ = o.view(sr, si, rng, rng, 4096, 4096, mrd, want_bytes=False)
    fused = o.view_contracted(sr, si, rng, rng, 4096, 4096, mrd)
    d = strict != fused
    db = numpy_quantise(strict, mrd) != numpy_quantise(fused, mrd)
    delta = np.abs(strict.astype(np.int64) - fused.astype(np.int64))[d]
    flips = int(((strict == 0) != (fused == 0)).sum())
    rows.append({"tile": [level, mrd, ir, ii], "pixels": int(strict.size), "counts_differ": int(d.sum()), "bytes_differ": int(db.sum()),
                 "in_set_membership_flips": flips, "max_count_delta": int(delta.max()) if delta.size else 0,
                 "median_count_delta": float(np.median(delta)) if delta.size else 0.0,
                 "escaped_pixels": int((strict > 0).sum())})
    print(rows[-1], f"{time.time() - t0:.1f}s", flush=True)
os.makedirs("profiles/r02", exist_ok=True)
json.dump({"what": "strict evaluation of WorkerCUDA.py:39-68 (this worker) vs the same loop with default CUDA FMA contraction "
                   "(fma(z0,z0,-z1*z1), fma(2*z0,z1,c1), fma(z0,z0,z1*z1)); 4096x4096 DataChunk tiles", "tiles": rows},
          open("profiles/r02/contraction_sensitivity.json", "w"), indent=1)

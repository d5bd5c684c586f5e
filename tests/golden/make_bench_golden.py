"""Golden outputs for bench.py's `output_verified` fields: sha256 of the int32 counts (little-endian, row-major) and the
totals of every workload the default `python bench.py` line times, computed by the CPU oracle (oracle/mandel_oracle.c -- the
strict scalar loop; the AVX-512 evaluation is used where it exists and is itself tested bit-identical to the scalar one,
tests/test_oracle.py::test_avx512_baseline_is_bit_identical_to_scalar_oracle).

    python tests/golden/make_bench_golden.py [name ...]      # ~15 CPU-minutes on 8 cores for all of them (cfg4's band: 10)

Writes / updates tests/golden/bench_outputs.json.  bench.py hashes the buffer its timed launches wrote (one D2H after the
timed region) and compares: the number it prints then belongs to the reference's output, inside the driver's own run.
The views are bench.py's WORKLOADS / EXTRA_CONFIGS (kept in step by tests/test_bench_contract.py)."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import COracle  # noqa: E402

# name: (start_r, start_i, range, width, height, mrd, precision, window (col0, row0, ncols, nrows) | None)
CASES = {
    "cfg2": (-2.0, -1.5, 3.0, 4096, 4096, 1000, "f64", None),
    "chunk_l1": (-2.0, -2.0, 4.0, 4096, 4096, 1000, "f64", None),
    "cfg5": (-2.0, -1.5, 3.0, 4096, 4096, 5000, "f64", None),
    "cfg3": (-0.743648, 0.131820, 1e-5, 8192, 8192, 10000, "f64", None),
    "cfg4_band": (-0.755, 0.10, 0.02, 16384, 16384, 50000, "f32", (0, 7680, 16384, 1024)),
}
OUT = os.path.join(ROOT, "tests", "golden", "bench_outputs.json")


def main():
    o = COracle()
    have = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in (sys.argv[1:] or list(CASES)):
        sr, si, rng, w, h, mrd, precision, window = CASES[name]
        t0 = time.time()
        if precision == "f64" and window is None and o.have_avx512():
            counts, total = o.view_avx512(sr, si, rng, rng, w, h, mrd)
            how = "mbo_view_avx512"
        else:
            counts, _, total = o.view(sr, si, rng, rng, w, h, mrd, window=window, want_bytes=False, precision=precision)
            how = "mbo_view_f32" if precision == "f32" else "mbo_view"
        counts = np.ascontiguousarray(counts, dtype="<i4")
        have[name] = {"view": [sr, si, rng, rng, w, h], "mrd": mrd, "precision": precision, "window": window,
                      "counts_sha256": hashlib.sha256(counts.tobytes()).hexdigest(),
                      "pixel_iterations": int(total), "never_pixels": int((counts == 0).sum()),
                      "oracle": how, "oracle_seconds": round(time.time() - t0, 1)}
        print(name, have[name], flush=True)
        with open(OUT, "w") as f:
            json.dump(have, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()

"""Small staged probe of a kernel variant against the oracle (run under `timeout` on the GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from distributedmandelbrot_amd import MandelbrotDevice, View
from oracle.oracle import COracle

kernel = sys.argv[1] if len(sys.argv) > 1 else "scan"
o = COracle()
dev = MandelbrotDevice(0)
cases = [(View(-2.0, -1.5, 3.0, 3.0, 16, 16), 50), (View(-2.0, -1.5, 3.0, 3.0, 64, 48), 100),
         (View(-2.0, -1.5, 3.0, 3.0, 77, 53), 256), (View(-0.2, -0.1, 0.2, 0.2, 64, 64), 100),
         (View(-2.0, -1.5, 3.0, 3.0, 512, 512), 256), (View(-2.0, -1.5, 3.0, 3.0, 1024, 1024), 1000),
         (View(-2.0, -1.5, 3.0, 3.0, 4096, 4096), 1000)]
for v, mrd in cases:
    t0 = time.time()
    c, b, st = dev.compute_view(v, mrd, kernel=kernel)
    oc, ob, tot = o.view(v.start_r, v.start_i, v.range_r, v.range_i, v.width, v.height, mrd)
    bad = int((c != oc).sum())
    print(f"{v.width}x{v.height} mrd {mrd}: mismatches {bad} bytes_ok {bool((b == ob).all())} kernel {st.kernel_ms:.3f} ms "
          f"{st.pixel_iterations / st.kernel_ms / 1e6:.1f} G/s wall {time.time() - t0:.2f}s", flush=True)
    if bad:
        idx = np.argwhere(c != oc)[:5]
        print("  first mismatches (row, col, got, want):", [(int(r), int(cc), int(c[r, cc]), int(oc[r, cc])) for r, cc in idx])
        break

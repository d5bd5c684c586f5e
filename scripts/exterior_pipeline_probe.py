"""What bounds the tile rate of the host-buffer pipeline on all-exterior tiles (3 in 4 tiles of a pyramid level)?
N exterior DataChunk tiles (level 16, mrd 1024: counts 1..4, byte 1 -- computed, uniform, not copied under MBK_LAZY_UNIFORM)
with k tiles in flight; prints the rate and the host time of submit / wait.  Run under
    rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --stats
to see the HIP calls and the GPU side (scripts/gpu_run.sh section `extprobe`).
    python scripts/exterior_pipeline_probe.py [tiles] [slots] [lazy 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # (run from /tmp under rocprofv3)
import numpy as np
from distributedmandelbrot_amd import MandelbrotDevice

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lazy = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
dev = MandelbrotDevice(0)
pins = [dev.pinned_empty((16777216,), np.uint8) for _ in range(k)]
# the Immediate tiles of level 16 that are NOT answered on the host (inside the circle's bounding box: counts 1..4, byte 1)
tiles = []
for ir in range(16):
    for ii in range(16):
        dev.submit_datachunk(0, 16, 1024, ir, ii, pins[0], lazy_uniform=True)
        st = dev.wait(0)
        if st.all_bytes_one and st.kernel_ms > 0:
            tiles.append((ir, ii))
for i in range(k):                                # first use of every slot (buffers, scratch)
    dev.submit_datachunk(i, 16, 1024, *tiles[i], pins[i], lazy_uniform=lazy)
for i in range(k):
    dev.wait(i)
ts = tw = 0.0
t0 = time.perf_counter()
for i in range(n + k):
    if i >= k:
        a = time.perf_counter()
        dev.wait((i - k) % k)
        tw += time.perf_counter() - a
    if i < n:
        a = time.perf_counter()
        dev.submit_datachunk(i % k, 16, 1024, *tiles[i % len(tiles)], pins[i % k], lazy_uniform=lazy)
        ts += time.perf_counter() - a
dt = time.perf_counter() - t0
print(f"{n} all-exterior tiles ({len(tiles)} distinct), {k} in flight, lazy={int(lazy)}: {n / dt:.0f} tiles/s = {dt / n * 1e6:.1f} us per tile; host: submit {ts / n * 1e6:.1f} us, wait {tw / n * 1e6:.1f} us")

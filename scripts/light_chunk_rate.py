"""One all-exterior DataChunk, over and over, the way the worker asks for it (bytes + statistics, uniform tiles not
copied off the GPU): kernel time, wall time per tile with two in flight.  Run under rocprofv3 --kernel-trace --stats for
the reduction's share.   python scripts/light_chunk_rate.py [level ir ii mrd] [opt=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from distributedmandelbrot_amd import MandelbrotDevice

nums = [int(a) for a in sys.argv[1:] if "=" not in a]
level, ir, ii, mrd = (nums + [4, 0, 0, 1000])[:4] if len(nums) >= 4 else (4, 0, 0, 1000)
dev = MandelbrotDevice(0)
for item in sys.argv[1:]:
    if "=" in item:
        k, _, v = item.partition("=")
        dev.set_option(k, int(v))
pins = [dev.pinned_empty((16777216,), np.uint8) for _ in range(2)]
for _ in range(200):                                   # warm-up and clock
    dev.datachunk(level, mrd, ir, ii, out_bytes=pins[0])
n, ks = 2000, []
t0 = time.perf_counter()
dev.submit_datachunk(0, level, mrd, ir, ii, pins[0], lazy_uniform=True)
for i in range(1, n + 1):
    if i < n:
        dev.submit_datachunk(i % 2, level, mrd, ir, ii, pins[i % 2], lazy_uniform=True)
    ks.append(dev.wait((i - 1) % 2).kernel_ms)
dt = time.perf_counter() - t0
print(f"DataChunk ({level},{ir},{ii}) mrd {mrd}: tile kernel ms median {np.median(ks):.4f} min {np.min(ks):.4f}; two in flight, lazy uniform: "
      f"{n/dt:.0f} tiles/s = {dt/n*1e6:.1f} us per tile wall")

"""End-to-end tile rate of the drop-in worker against the Python stand-in Distributer (loopback TCP).
Run on the GPU box:  python scripts/worker_e2e.py [level] [mrd] [feeders]"""
import sys, time, threading
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from fake_distributer import FakeDistributer
from distributedmandelbrot_amd import worker, MandelbrotDevice
from distributedmandelbrot_amd.device import device_count

level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
mrd = int(sys.argv[2]) if len(sys.argv) > 2 else 256
feeders = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ngpu = device_count()
print(f"GPUs visible: {ngpu}; level {level} ({level*level} tiles), mrd {mrd}, feeders {feeders}")
# kernel-only and kernel+D2H per tile
dev = MandelbrotDevice(0)
pin = dev.pinned_empty((worker.CHUNK_BYTES,), np.uint8)
ks, ds = [], []
t0 = time.perf_counter()
for ir in range(level):
    for ii in range(level):
        _, _, st = dev.datachunk(level, mrd, ir, ii, out_bytes=pin)
        ks.append(st.kernel_ms); ds.append(st.d2h_ms)
t_compute = time.perf_counter() - t0
print(f"compute only (pinned D2H): {level*level/t_compute:.1f} tiles/s; kernel ms mean {np.mean(ks):.3f} max {np.max(ks):.3f}; d2h ms mean {np.mean(ds):.3f}")
dev.close()
with FakeDistributer([(level, mrd)]) as srv:
    t0 = time.perf_counter()
    done = worker.run_farm("127.0.0.1", srv.port, devices=[0] * feeders, log=lambda *a: None)
    assert srv.wait_completed(level * level, timeout=120)
    dt = time.perf_counter() - t0
print(f"through the wire protocol (single-threaded stand-in Distributer, loopback): {level*level/dt:.1f} tiles/s ({dt/level/level*1e3:.1f} ms/tile), per feeder {done}")

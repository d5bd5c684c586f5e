"""End-to-end tile rate of the drop-in worker against the threaded stand-in Distributer (loopback TCP):
the reference's serial loop (do_workload_single) vs the pipelined feeder (run_pipelined: lease / compute / send
overlapped, two tiles in flight on the GPU) vs the same loop in native code (run_native = mbk_worker_run).  Run on the GPU box:
    python scripts/worker_e2e.py [level] [mrd] [senders]"""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from distributedmandelbrot_amd import worker, MandelbrotDevice
from distributedmandelbrot_amd.server import Distributer

level = int(sys.argv[1]) if len(sys.argv) > 1 else 12
mrd = int(sys.argv[2]) if len(sys.argv) > 2 else 256
senders = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = level * level
QUIET = lambda *a: None


def settle(dist, k, timeout=120.0):
    end = time.time() + timeout
    while dist.received < k and time.time() < end:
        time.sleep(0.002)
    return dist.received >= k


dev = MandelbrotDevice(0)
pin = dev.pinned_empty((worker.CHUNK_BYTES,), np.uint8)
ks, ds = [], []
t0 = time.perf_counter()
for ir in range(level):
    for ii in range(level):
        _, _, st = dev.datachunk(level, mrd, ir, ii, out_bytes=pin)
        ks.append(st.kernel_ms); ds.append(st.d2h_ms)
t_compute = time.perf_counter() - t0
print(f"level {level} ({n} tiles), mrd {mrd}: compute only (pinned D2H) {n / t_compute:.1f} tiles/s; kernel ms mean {np.mean(ks):.3f} "
      f"median {np.median(ks):.3f} max {np.max(ks):.3f}; d2h ms mean {np.mean(ds):.3f}")

with Distributer([(level, mrd)]) as dist:          # serial reference-shaped loop on the same device
    def compute(lv, m, ir, ii):
        out, _, _ = dev.datachunk(lv, m, ir, ii, out_bytes=pin)
        return out
    t0 = time.perf_counter()
    k = 0
    while worker.do_workload_single("127.0.0.1", dist.port, compute=compute, log=QUIET):
        k += 1
    assert settle(dist, n)
    dt = time.perf_counter() - t0
print(f"serial worker loop (WorkerCUDA.py:111-176 shape): {n / dt:.1f} tiles/s ({dt / n * 1e3:.2f} ms/tile)")

for s in sorted({1, 2, senders}):
    with Distributer([(level, mrd)]) as dist:
        t0 = time.perf_counter()
        done = worker.run_pipelined("127.0.0.1", dist.port, device=dev, log=QUIET, senders=s)
        assert done == n and settle(dist, n)
        dt = time.perf_counter() - t0
    print(f"pipelined worker, {s} sender thread(s): {n / dt:.1f} tiles/s ({dt / n * 1e3:.2f} ms/tile); stats {dict(worker.stats)}")
for s in sorted({2, 4, 8, senders}):
    with Distributer([(level, mrd)]) as dist:
        t0 = time.perf_counter()
        done = worker.run_native("127.0.0.1", dist.port, device=dev, log=QUIET, senders=s)
        assert done == n and settle(dist, n)
        dt = time.perf_counter() - t0
    print(f"native feeder (mbk_worker_run), {s} sender thread(s): {n / dt:.1f} tiles/s ({dt / n * 1e3:.2f} ms/tile)")
# ... and against a native sink (scripts/sink_distributer.cpp: same protocol, one C++ thread per connection, payload
# read and dropped): what the feeder sustains when the server is not the bottleneck
import os, subprocess, tempfile
exe = os.path.join(tempfile.gettempdir(), "mbk_sink_distributer")
subprocess.check_call(["g++", "-O2", "-pthread", "-o", exe, os.path.join(os.path.dirname(os.path.abspath(__file__)), "sink_distributer.cpp")])
for s in sorted({4, 8, 16, senders}):
    srv = subprocess.Popen([exe, str(level), str(mrd)], stdout=subprocess.PIPE, text=True)
    port = int(srv.stdout.readline().split()[1])
    t0 = time.perf_counter()
    done = worker.run_native("127.0.0.1", port, device=dev, log=QUIET, senders=s)
    line = srv.stdout.readline().strip()
    dt = time.perf_counter() - t0
    srv.wait(timeout=10)
    assert done == n and line.startswith(f"DONE {n} "), (done, line)
    print(f"native feeder -> native sink, {s} sender thread(s): {n / dt:.1f} tiles/s ({dt / n * 1e3:.2f} ms/tile)")
dev.close()

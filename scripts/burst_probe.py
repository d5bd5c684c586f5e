"""How long does a burst of cfg2 launches take to reach its steady rate, by the idle gap before it?  (round 6: the driver's
20-step timed region read 2-3 % below the >= 1 s `sustained` leg of the same line.)
    python scripts/burst_probe.py [cycle_detect]
Per gap: a 300 ms ramp (launch + sync), a host sleep of `gap`, then 60 launches back to back with a pair of events per launch
(which costs ~2 % by itself: compare shapes, not levels), and the same burst timed as ONE region of 20 launches five times in a
row (5 untimed launches + a sync before each, as bench.py's warm-up does)."""
import sys
import time

sys.path.insert(0, ".")
import torch  # noqa: E402

from distributedmandelbrot_amd import MandelbrotDevice, View  # noqa: E402

cyc = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = MandelbrotDevice(0)
dev.set_option("cycle_detect", cyc)
view, mrd = View(-2.0, -1.5, 3.0, 3.0, 4096, 4096), 1000
buf = torch.empty(4096 * 4096, dtype=torch.int32, device="cuda:0")
s = torch.cuda.current_stream()


def launch():
    dev.launch_view(view, mrd, d_counts=buf.data_ptr(), stream=s.cuda_stream)


def ramp(ms, synced=True):
    t = time.perf_counter()
    while (time.perf_counter() - t) * 1e3 < ms:
        launch()
        if synced:
            torch.cuda.synchronize()
    torch.cuda.synchronize()


for gap_ms in (0.0, 0.2, 2.0, 20.0, 200.0):
    ramp(300)
    time.sleep(gap_ms / 1e3)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
    for a, b in ev:
        a.record(s)
        launch()
        b.record(s)
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    print(f"gap {gap_ms:6.1f} ms: per-launch ms " + " ".join(f"{x:.3f}" for x in ms[:12]) + f" ... mean of 13-60: {sum(ms[12:]) / 48:.4f}")

for name, synced, ramp_ms in (("synced ramp 150 ms (bench.py)", True, 150), ("synced ramp 600 ms", True, 600), ("back-to-back ramp 150 ms", False, 150), ("back-to-back ramp 600 ms", False, 600)):
    out = []
    for rep in range(6):
        time.sleep(0.3)                      # the host work between legs: the GPU idles
        ramp(ramp_ms, synced)
        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20):
            launch()
        e1.record(s)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 20)
    print(f"{name:32s}: 20-launch region, ms per launch: " + " ".join(f"{x:.4f}" for x in out))
# a long region for reference
ramp(300)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s)
for _ in range(1500):
    launch()
e1.record(s)
torch.cuda.synchronize()
print(f"1500 launches back to back: {e0.elapsed_time(e1) / 1500:.4f} ms per launch")
dev.close()

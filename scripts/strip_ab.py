"""MBK_OPT_SCAN_STRIP on / off on one box (run on the GPU box): the all-exterior DataChunk (4,0,0) for every output set, and
the GPU side of a whole pyramid level (256 tiles of level 16, bytes only, no copies), launches back to back on one stream,
timed with one event pair per leg, legs interleaved.   python scripts/strip_ab.py [reps]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from distributedmandelbrot_amd import MandelbrotDevice, View

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
N = 4096
d_counts = torch.empty((N, N), dtype=torch.int32, device="cuda")
d_bytes = torch.empty((N, N), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
devs = {}
for strip in (0, 1):
    devs[strip] = MandelbrotDevice(0)
    devs[strip].set_option("scan_strip", strip)


def leg(dev, views, mrd, want, k, precision="f64"):
    kw = {}
    if want in ("counts", "both"):
        kw["d_counts"] = d_counts.data_ptr()
    if want in ("bytes", "both"):
        kw["d_bytes"] = d_bytes.data_ptr()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        for v in views:
            dev.launch_view(v, mrd, stream=stream, precision=precision, **kw)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (k * len(views))   # us per tile


ext = [View(-2.0, -2.0, 1.0, 1.0, N, N)]
level16 = [View(-2.0 + 0.25 * i, -2.0 + 0.25 * j, 0.25, 0.25, N, N) for j in range(16) for i in range(16)]
# clock ramp
for _ in range(3):
    leg(devs[0], ext, 1024, "counts", 300)
print("us per tile, median of", reps, "(strip 0 / strip 1)")
for name, views, k in (("exterior DataChunk (4,0,0)", ext, 300), ("level 16, 256 tiles, GPU side", level16, 1)):
    for precision in ("f64", "f32"):
        for want in ("counts", "bytes", "both"):
            if views is level16 and (want != "bytes" or precision != "f64"):
                continue
            t = {0: [], 1: []}
            for _ in range(reps):
                for strip in (0, 1):
                    t[strip].append(leg(devs[strip], views, 1024, want, k, precision))
            a, b = float(np.median(t[0])), float(np.median(t[1]))
            nbytes = N * N * ({"counts": 4, "bytes": 1, "both": 5}[want])
            print(f"{name:32s} {precision} {want:6s} {a:9.2f} {b:9.2f}  x{a / b:.3f}"
                  + (f"   {nbytes / b / 1e6:.2f} TB/s with strips" if views is ext else f"   level: {a * 256 / 1e3:.2f} -> {b * 256 / 1e3:.2f} ms"))
# same output either way
for strip in (0, 1):
    devs[strip].launch_view(ext[0], 1024, d_counts=d_counts.data_ptr(), d_bytes=d_bytes.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    print("strip", strip, "checksum counts", int(d_counts.sum(dtype=torch.int64)), "bytes", int(d_bytes.sum(dtype=torch.int64)))

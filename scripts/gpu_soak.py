"""One-off randomized parity soak: random views x kernels x precisions against the CPU oracle.
    timeout 900 python scripts/gpu_soak.py [seconds] [seed]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from distributedmandelbrot_amd import MandelbrotDevice, View
from oracle.oracle import COracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
o = COracle()
dev = MandelbrotDevice(0)
combos = [("group", "f64"), ("asm", "f64"), ("refill", "f64"), ("simple", "f64"), ("group", "f32"), ("asm", "f32")]
t0 = time.time(); n = 0; px = 0
while time.time() - t0 < budget:
    kind = rs.randint(0, 6)
    if kind == 0:
        cr, ci = rs.uniform(-2.1, 2.1), rs.uniform(-2.1, 2.1)
    elif kind == 1:
        th = rs.uniform(0, 2 * np.pi); r = 2 + rs.uniform(-1e-8, 1e-8)
        cr, ci = r * np.cos(th), r * np.sin(th)
    elif kind == 2:
        th = rs.uniform(0, 2 * np.pi)
        cr, ci = 0.5 * np.cos(th) - 0.25 * np.cos(2 * th), 0.5 * np.sin(th) - 0.25 * np.sin(2 * th)
    elif kind == 3:
        cr, ci = rs.uniform(-2, 0.3), rs.choice([0.0, 1e-300, -1e-310, 1e-17])
    elif kind == 4:
        cr, ci = -0.743643 + rs.uniform(-1e-4, 1e-4), 0.131825 + rs.uniform(-1e-4, 1e-4)
    else:
        th = rs.uniform(0, 2 * np.pi); cr, ci = -1 + 0.25 * np.cos(th), 0.25 * np.sin(th)
    span_r = 10.0 ** rs.uniform(-11, 0.5); span_i = span_r * rs.uniform(0.2, 5.0)
    w, h = int(rs.randint(1, 200)), int(rs.randint(1, 200))
    mrd = int(rs.choice([2, 3, 8, 9, 10, 16, 17, 18, 24, 25, 26, 33, 100, 257, 1000, 4000]))
    view = View(cr - span_r / 2, ci - span_i / 2, span_r, span_i, w, h)
    window = None
    if w > 3 and h > 3 and rs.rand() < 0.3:
        c0, r0 = int(rs.randint(0, w - 1)), int(rs.randint(0, h - 1))
        window = (c0, r0, int(rs.randint(1, w - c0 + 1)), int(rs.randint(1, h - r0 + 1)))
    kernel, prec = combos[rs.randint(0, len(combos))]
    c, b, st = dev.compute_view(view, mrd, window=window, kernel=kernel, precision=prec)
    oc, ob, total = o.view(view.start_r, view.start_i, view.range_r, view.range_i, w, h, mrd, window=window, precision=prec)
    if not (np.array_equal(c, oc) and np.array_equal(b, ob) and st.pixel_iterations == total):
        print("MISMATCH", kernel, prec, view, mrd, window, int((c != oc).sum()), flush=True)
        idx = np.argwhere(c != oc)[:5]
        print([(int(r), int(cc), int(c[r, cc]), int(oc[r, cc])) for r, cc in idx])
        sys.exit(1)
    n += 1; px += c.size
print(f"soak ok: {n} random views, {px/1e6:.1f} Mpixel, {time.time()-t0:.0f} s, seed {seed}")

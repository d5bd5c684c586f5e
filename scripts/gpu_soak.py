"""Randomized parity soak: random views x kernels x precisions x tuning options x output sets against the CPU
oracle (run on the GPU box).   timeout 900 python scripts/gpu_soak.py [seconds] [seed]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from distributedmandelbrot_amd import MandelbrotDevice, View
from oracle.oracle import COracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
o = COracle()
combos = [("scan", "f64"), ("default", "f64"), ("group", "f64"), ("asm", "f64"), ("refill", "f64"), ("simple", "f64"),
          ("scan", "f32"), ("group", "f32"), ("asm", "f32")]
OPTION_CHOICES = {"scan_waves": [1, 2, 8], "scan_xcd_map": [0, 1], "scan_col_period": [0, 1, 4], "group_steps": [4, 8, 16, 32], "scan_inline": [0, 1],
                  "exact_steps": [0, 3, 8, 20], "order": [0, 1, 2, 3], "xcd_balance": [0, 1, 2], "units_min_light": [0, 32768], "heavy_share": [0, 655, 65536], "waves_per_wg": [1, 2, 4],
                  "cycle_detect": [0, 1], "probe_mid": [2, 6, 65537], "prepass_overlap": [0, 1, 2], "probe_steps": [2, 32, 200], "exact_long": [0, 4, 8],
                  "m_late": [0, 4, 8, 12], "h_settled": [0, 5, 6, 9], "classify_wg": [64, 256, 1024], "scan_strip": [0, 1], "cycle_window": [0, 6, 32, 4096],
                  "spill_first": [0, 32, 64, 256], "spill_lanes": [1, 7, 16, 32], "spill_min_mrd": [2, 100], "spill_min_blocks": [0, 0, 19], "spill_cyc_shift": [0, 5, 31]}
t0 = time.time(); n = 0; px = 0
dev = None
while time.time() - t0 < budget:
    if dev is None or n % 25 == 0:          # a fresh ctx with a random option set every 25 views
        if dev is not None:
            dev.close()
        dev = MandelbrotDevice(0)
        opts = {k: int(rs.choice(v)) for k, v in OPTION_CHOICES.items() if rs.rand() < 0.5}
        for k, v in opts.items():
            dev.set_option(k, v)
    kind = rs.randint(0, 8)
    needle = kind == 7
    if needle:   # the antenna on the real axis inside a tall window: often invisible to the 16 x 16 host probe (finish-in-place light pass)
        cr, ci = rs.uniform(-2.0, -1.45), rs.uniform(-0.02, 0.02)
    elif kind == 0:
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
    elif kind == 5:
        cr, ci = rs.uniform(-3, 3), rs.uniform(-3, 3)          # mostly far exterior: the light path
    else:
        th = rs.uniform(0, 2 * np.pi); cr, ci = -1 + 0.25 * np.cos(th), 0.25 * np.sin(th)
    span_r = 10.0 ** rs.uniform(-11, 0.7); span_i = span_r * rs.uniform(0.2, 5.0)
    big = rs.rand() < 0.25                                       # enough blocks for several sweeps of the light pass
    if needle:
        span_r, span_i, big = rs.uniform(0.05, 0.4), rs.uniform(0.8, 3.2), True
    w, h = (int(rs.randint(200, 1400)), int(rs.randint(200, 1100))) if big else (int(rs.randint(1, 200)), int(rs.randint(1, 200)))
    mrd = int(rs.choice([2, 3, 4, 5, 6, 8, 9, 10, 16, 17, 18, 24, 25, 26, 33, 41, 57, 100, 257, 1000] + ([] if big else [4000, 20000])))
    if kind in (2, 4, 6) and rs.rand() < 0.35:      # what the units kernel serves: >= 16 384 blocks, block
        w, h = 64 * int(rs.randint(16, 22)), int(rs.randint(1024, 1300))   # columns a multiple of 8, on the boundary of the set
        mrd = int(rs.choice([150, 300, 700, 2500]))
        span_r = 10.0 ** rs.uniform(-6, -0.5); span_i = span_r * rs.uniform(0.5, 2.0)
    view = View(cr - span_r / 2, ci - span_i / 2, span_r, span_i, w, h)
    window = None
    if w > 3 and h > 3 and rs.rand() < 0.3:
        c0, r0 = int(rs.randint(0, w - 1)), int(rs.randint(0, h - 1))
        window = (c0, r0, int(rs.randint(1, w - c0 + 1)), int(rs.randint(1, h - r0 + 1)))
    kernel, prec = combos[rs.randint(0, len(combos))]
    oc, ob, total = o.view(view.start_r, view.start_i, view.range_r, view.range_i, w, h, mrd, window=window, precision=prec)
    outs = rs.randint(0, 5) if kernel != "refill" else 0       # 0: host API (counts + bytes); 1: bytes only; 2: counts only;
                                                               # 3 / 4: host API with bytes only / counts only (statistics may be fused)
    if outs == 0 or outs >= 3:
        c, b, st = dev.compute_view(view, mrd, window=window, kernel=kernel, precision=prec, want_counts=outs != 3, want_bytes=outs != 4)
        ok = (c is None or np.array_equal(c, oc)) and (b is None or np.array_equal(b, ob)) and st.pixel_iterations == total \
            and st.never_pixels == int((oc == 0).sum())
        if b is not None:
            ok = ok and st.rle_runs == 1 + int((ob.ravel()[1:] != ob.ravel()[:-1]).sum()) and st.all_bytes_zero == bool((ob == 0).all())
        if c is None:
            c = b
    else:                                                        # device-pointer launch with a single output
        npx = oc.size
        t = torch.full((npx,), 77, dtype=torch.uint8 if outs == 1 else torch.int32, device="cuda:0")
        s = torch.cuda.current_stream().cuda_stream
        dev.launch_view(view, mrd, window=window, kernel=kernel, precision=prec, stream=s,
                        **({"d_bytes": t.data_ptr()} if outs == 1 else {"d_counts": t.data_ptr()}))
        torch.cuda.synchronize()
        got = t.cpu().numpy().reshape(oc.shape)
        c = got
        ok = np.array_equal(got, ob if outs == 1 else oc)
    if not ok:
        print("MISMATCH", kernel, prec, opts, "outs", outs, view, mrd, window, flush=True)
        ref = ob if outs in (1, 3) else oc
        idx = np.argwhere(c != ref)[:5]
        print([(int(r), int(cc), int(c[r, cc]), int(ref[r, cc])) for r, cc in idx])
        sys.exit(1)
    n += 1; px += oc.size
print(f"soak ok: {n} random views, {px/1e6:.1f} Mpixel, {time.time()-t0:.0f} s, seed {seed}")

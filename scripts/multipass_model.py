"""CPU model: what would re-packing the surviving pixels into full waves at a few step checkpoints buy (a multi-pass kernel: pass k runs
every wave to step P_k, survivors are handed on through a compact list in block order and packed 64 to a wave)?  Wave-steps on the exact counts
of the view (oracle), against the one-pass 8x8-block scheme.  No overheads modelled (state hand-over, scattered stores, extra launches).
    python scripts/multipass_model.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle.oracle import COracle
o = COracle()
def model(name, view, N, mrd, passes_list):
    t=time.time()
    sr, si, rr, ri = view
    c, _ = o.view_avx512(sr, si, rr, ri, N, N, mrd) if o.have_avx512() else o.view(sr,si,rr,ri,N,N,mrd,want_bytes=False)[:2]
    T = mrd - 1
    nb = N // 8
    steps = np.where(c == 0, T, c).astype(np.int64)
    B = steps.reshape(nb, 8, nb, 8).transpose(0, 2, 1, 3).reshape(-1, 64)   # block-major pixel order
    ideal = B.sum() / 64.0
    single = B.max(1).sum()
    print(f"{name}: oracle {time.time()-t:.0f}s  ideal wave-steps {ideal/1e6:.1f} M; one pass (8x8 blocks) {single/1e6:.1f} M  lane activity {ideal/single:.3f}")
    flat = B.reshape(-1)          # list order = block order, lane order inside
    for passes in passes_list:
        bounds = [0] + list(passes) + [T]
        total = 0.0; cur = flat; extra_px = 0
        for k in range(len(bounds) - 1):
            lo, hi = bounds[k], bounds[k + 1]
            if k == 0:
                W = cur.reshape(-1, 64)
            else:
                n = len(cur); pad = (-n) % 64
                W = np.concatenate([cur, np.full(pad, lo, np.int64)]).reshape(-1, 64)
                extra_px += n
            run = np.minimum(W, hi).max(1) - lo
            total += run.sum()
            cur = cur[cur > hi]
        print(f"   passes at {passes}: wave-steps {total/1e6:.1f} M ({total/single:.3f} of one pass; activity {ideal/total:.3f}); pixels handed on {extra_px/1e6:.1f} M")
model("cfg3", (-0.743648, 0.131820, 1e-5, 1e-5), 4096, 10000, [(512,), (256, 2048), (128, 512, 2048), (64, 256, 1024, 4096)])
model("cfg2", (-2.0, -1.5, 3.0, 3.0), 4096, 1000, [(64,), (32, 256)])

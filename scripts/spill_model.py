"""CPU model of a GLOBAL re-pack ("spill", round 6): a wave of the one-wave-per-8x8-block kernel that reaches step S with only
k <= T of its 64 lanes alive writes the live lanes' state (zr, zi, pixel id) to a list in HBM and ends; a second pass runs the
listed lanes 64 to a wave from step S on (escape_steps_tail resumes from any state).  Unlike the in-workgroup re-pack of round 4
(scripts/repack_model.py: 16x16 regions, 4-wave workgroups) the packing is global -- a wave of pass 2 holds lanes of any
blocks -- and the workgroups stay single waves.  Wave-steps on the exact counts (oracle), strict and cycle-test schedules.
    python scripts/spill_model.py [--views cfg2,chunk_l1,cfg3]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle.oracle import COracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--views", default="cfg2,chunk_l1,cfg3")
ap.add_argument("--spill-cost", type=float, default=6.0, help="wave-steps a spill costs its wave (stores + atomics ~ 40 instructions)")
ap.add_argument("--load-cost", type=float, default=6.0, help="wave-steps a pass-2 wave spends loading its lanes")
args = ap.parse_args()
o = COracle()
VIEWS = {"cfg2": ((-2.0, -1.5, 3.0, 3.0), 4096, 1000), "chunk_l1": ((-2.0, -2.0, 4.0, 4.0), 4096, 1000),
         "cfg3": ((-0.743648, 0.131820, 1e-5, 1e-5), 4096, 10000)}


def blocks(x, N):
    nb = N // 8
    return x.reshape(nb, 8, nb, 8).transpose(0, 2, 1, 3).reshape(nb * nb, 64)


def model(name, steps, resumed, N, total, S_list, T_list):
    """steps[p]: steps pixel p executes in one go; resumed[S][p]: steps it executes in all when it is re-started at step S"""
    B = blocks(steps, N).astype(np.int64)
    ideal = B.sum() / 64.0
    single = B.max(1).sum()
    print(f"{name}: ideal {ideal / 1e6:.2f} M wave-steps; one wave per block {single / 1e6:.2f} M (lane activity {ideal / single:.3f})")
    for S in S_list:
        R = blocks(resumed[S], N).astype(np.int64) if resumed is not None else B
        alive = B > S
        k = alive.sum(1)
        for T in T_list:
            spill = (k > 0) & (k <= T)
            p1 = np.where(spill, S, B.max(1)).sum() + spill.sum() * args.spill_cost
            lanes = (R[spill] - S)[alive[spill]]            # remaining steps of the spilled lanes, in list (image) order
            nw = (len(lanes) + 63) // 64
            pad = np.zeros(nw * 64, np.int64)
            pad[:len(lanes)] = lanes
            p2 = pad.reshape(nw, 64).max(1).sum() + nw * args.load_cost
            tot = p1 + p2
            print(f"   S {S:5d} T {T:2d}: {spill.sum():6d} blocks spill {len(lanes):7d} lanes -> {nw:5d} waves; pass 1 {p1 / 1e6:7.2f} M + pass 2 {p2 / 1e6:6.2f} M"
                  f" = {tot / 1e6:7.2f} M = {tot / single:.4f} of one-wave-per-block (activity {ideal / tot:.3f})")


for name in args.views.split(","):
    view, N, mrd = VIEWS[name]
    t = time.time()
    c = o.view_avx512(*view, N, N, mrd)[0] if o.have_avx512() else o.view(*view, N, N, mrd, want_bytes=False)[0]
    total = mrd - 1
    strict = np.where(c == 0, total, c)
    S_list = [32, 64, 128, 256] if mrd <= 1000 else [256, 512, 1024, 2048]
    print(f"-- {name} ({N}^2, mrd {mrd}; oracle {time.time() - t:.0f} s)")
    model(name + " strict", strict, None, N, total, S_list, [16, 32, 48, 63])
    _, ex = o.view_cycle(*view, N, N, mrd, first=8, check=8, window_cap=32)
    cyc = np.where(c == 0, ex, c)
    resumed = {}
    for S in S_list:   # the cycle test re-started at step S (new reference state, window 1): what pass 2 runs
        _, ex2 = o.view_cycle(*view, N, N, mrd, first=S, check=8, window_cap=32)
        resumed[S] = np.where(c == 0, np.maximum(ex2, S), c)
    model(name + " cycle test", cyc, resumed, N, total, S_list, [16, 32, 48, 63])

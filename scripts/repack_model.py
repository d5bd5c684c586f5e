"""CPU model of an IN-WORKGROUP re-pack: a 4-wave workgroup owns a 16x16-pixel region (four 8x8
blocks); at a few step checkpoints the surviving pixels of the region are compacted through LDS into as few waves as
hold them (state = zr, zi, cr, ci + pixel id per lane; results scattered from the LDS-held ids at the end), and waves
left without pixels exit.  Wave-steps on the exact counts of the view (oracle), for the strict schedule and for the
cycle-test schedule (executed steps from oracle.view_cycle), against the one-wave-per-8x8-block scheme -- and then
priced with the overheads measured on the chip:
  * a compaction costs every wave of the workgroup ~24 VALU/LDS instructions + a barrier (--repack-instr, in units of
    wave-steps: 24 / 6.125 = ~4 wave-steps per wave and checkpoint);
  * 4-wave workgroups cost dispatch flexibility: the same kernel with MBK_OPT_WAVES_PER_WG = 4 against 1, measured on the
    same box (--wg4-penalty, default from profiles/r04: see NOTES).
    python scripts/repack_model.py [--wg4-penalty-cfg3 X --wg4-penalty-cfg2 Y]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle.oracle import COracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--repack-steps", type=float, default=4.0, help="cost of one compaction per wave of the workgroup, in wave-steps")
ap.add_argument("--wg4-penalty-cfg3", type=float, default=None, help="measured time ratio waves_per_wg=4 / 1 on cfg3 (strict)")
ap.add_argument("--wg4-penalty-cfg2", type=float, default=None, help="the same on cfg2 with the cycle test")
args = ap.parse_args()
o = COracle()


def regions(x, N):
    """[region][256] with the four 8x8 blocks of a 16x16 region as four consecutive groups of 64"""
    nb = N // 16
    r = x.reshape(nb, 2, 8, nb, 2, 8).transpose(0, 3, 1, 4, 2, 5)     # [ry, rx, by, bx, y, x]
    return r.reshape(nb * nb, 4, 64)


def model(name, steps, N, T, checkpoints_list, penalty):
    R = regions(steps, N)                       # steps each pixel executes (strict: count or T; cycle leg: executed)
    ideal = R.sum() / 64.0
    single = R.max(2).sum()                     # one wave per 8x8 block, lock-step
    print(f"{name}: ideal {ideal / 1e6:.1f} M wave-steps; one wave per 8x8 block {single / 1e6:.1f} M (lane activity {ideal / single:.3f})")
    for cps in checkpoints_list:
        bounds = [0] + list(cps) + [T]
        total = over = 0.0
        waves = R.astype(np.int64)                 # [region, wave, lane]: steps the lane's pixel executes; -1 = empty lane
        for k in range(len(bounds) - 1):
            lo, hi = bounds[k], bounds[k + 1]
            total += (np.minimum(waves.max(2), hi) - lo).clip(min=0).sum()      # lock-step: a wave runs as long as its slowest lane
            if hi < T:
                alive = waves > hi
                alive_waves = alive.any(2).sum(1)                                # waves that reach the checkpoint with a live lane
                over += float(alive_waves.sum()) * args.repack_steps
                # order-preserving compaction (prefix sum over the ballots): survivors keep their order, 64 to a wave
                flat = np.where(alive, waves, -1).reshape(len(waves), 256)
                idx = np.argsort(~alive.reshape(len(waves), 256), axis=1, kind="stable")
                waves = np.take_along_axis(flat, idx, axis=1).reshape(len(waves), 4, 64)
        act = ideal / total
        line = (f"   checkpoints {str(cps):18s} wave-steps {total / 1e6:7.1f} M = {total / single:.3f} of one-wave-per-block (activity {act:.3f}); "
                f"+ compaction {over / 1e6:5.1f} M -> {(total + over) / single:.3f}")
        if penalty:
            line += f"; x 4-wave-workgroup penalty {penalty:.3f} -> {(total + over) / single * penalty:.3f}"
        print(line)


def run(name, view, N, mrd, cps_strict, cps_cycle, penalty_strict, penalty_cycle):
    t = time.time()
    sr, si, rr, ri = view
    c = o.view_avx512(sr, si, rr, ri, N, N, mrd)[0] if o.have_avx512() else o.view(sr, si, rr, ri, N, N, mrd, want_bytes=False)[0]
    T = mrd - 1
    strict = np.where(c == 0, T, c).astype(np.int64)
    print(f"-- {name} ({N}^2 sample of the view, mrd {mrd}; oracle {time.time() - t:.0f} s)")
    model(name + " strict", strict, N, T, cps_strict, penalty_strict)
    _, ex = o.view_cycle(sr, si, rr, ri, N, N, mrd, first=8, check=8)
    cyc = np.where(c == 0, ex, c).astype(np.int64)
    model(name + " cycle test", cyc, N, T, cps_cycle, penalty_cycle)


run("cfg3", (-0.743648, 0.131820, 1e-5, 1e-5), 4096, 10000, [(512,), (256, 2048), (128, 512, 2048)], [(512,), (256, 2048)],
    args.wg4_penalty_cfg3, args.wg4_penalty_cfg3)
run("cfg2", (-2.0, -1.5, 3.0, 3.0), 4096, 1000, [(64,), (32, 256)], [(64,), (64, 256), (32, 128, 512)],
    args.wg4_penalty_cfg2, args.wg4_penalty_cfg2)

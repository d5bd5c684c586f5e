"""Instruction-count model of BASELINE cfg2 under the one-wave-per-8x8-block kernels (CPU only).

    python scripts/cycle_model.py [--mrd 1000] [--size 4096]

Answers two questions with the exact counts of the tile (oracle, AVX-512 when present):
  1. where do the VALU instructions of a strict launch go (PMC r3: 305.5 M per cfg2 launch; r2: 311.9 M)?  Per
     block: a fixed overhead, the per-step prologue (8 VALU per step), 8- or 16-step groups (6G + 2), and the exact
     replay (9 per replayed step) -- `--replay spot`: once per group in which a lane tripped the test (rounds 1-2);
     `--replay deferred` (default, round 3): a trip costs 1 or 5 instructions and ONE fix-up per block replays
     all tripped lanes together -- against the lock-step floor (longest lane x 6.125) and the ideal
     (pixel-iterations x 6.125 / 64);
  2. how many wave-steps does the cycle test (MBK_OPT_CYCLE_DETECT, mbk_loops.inc) remove?  oracle.view_cycle
     models it per pixel (first 8 steps unchecked, then a bitwise state compare every --check steps -- 8 since
     round 3, 16 before --, Brent windows); a wave runs until its last lane escaped or was retired.
The numbers in DESIGN.md section 4 come from this script."""
import argparse
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle.oracle import COracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mrd", type=int, default=1000)
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--view", type=float, nargs=4, default=[-2.0, -1.5, 3.0, 3.0])
ap.add_argument("--fixed", type=float, default=40.0, help="VALU instructions per block outside the loops (calibrated on PMC)")
ap.add_argument("--replay", default="deferred", choices=["deferred", "spot"])
ap.add_argument("--check", type=int, default=8, help="steps between two bitwise state compares of the cycle test")
ap.add_argument("--window-cap", type=int, default=32, help="MBK_OPT_CYCLE_WINDOW: the saved state's window grows by a quarter below this many checks, doubles from there on (0 = always doubles, rounds 2-4)")
args = ap.parse_args()
o = COracle()
N, T, E = args.size, args.mrd - 1, 8
sr, si, rr, ri = args.view
if o.have_avx512():
    c, _ = o.view_avx512(sr, si, rr, ri, N, N, args.mrd)
else:
    c, _, _ = o.view(sr, si, rr, ri, N, N, args.mrd, want_bytes=False)
nb = N // 8


def blocks(x):
    return x.reshape(nb, 8, nb, 8).transpose(0, 2, 1, 3).reshape(-1, 64)


B = blocks(c).astype(np.int64)
e = np.where(B == 0, 10 ** 9, B)
center = c[4::8, 4::8].reshape(-1)
heavy = (center == 0) | (center >= 32)            # the classify probe: centre pixel alive after 32 steps
steps = np.where(B == 0, T, B)
pixit = int(steps.sum())
last = np.minimum(e.max(1), T)
print(f"pixel-iterations {pixit/1e9:.3f} G; ideal {pixit*6.125/64/1e6:.1f} M VALU; lock-step floor {last.sum()*6.125/1e6:.1f} M "
      f"({last.sum()/1e6:.2f} M wave-steps); blocks {len(B)}, probe-heavy {int(heavy.sum())}")


def grouped_cost(last_step):
    """VALU instructions of prologue + grouped loops per block when lane j leaves at step min(e_j, last_step_j)."""
    ee = np.minimum(e, last_step)                     # a retired lane simply leaves EXEC at its retirement step
    run = np.minimum(ee.max(1), T)
    cost = 8 * np.minimum(run, E)
    for s in range(1, E + 1):
        cost += (e == s).any(1)
    for flag, G in ((True, 16), (False, 8)):
        idx = np.where((heavy == flag) & (run > E))[0]
        x, esc = ee[idx], e[idx]
        n = np.full(len(idx), E)
        alive = x > E
        act = alive.any(1)
        co = np.zeros(len(idx), dtype=np.int64)
        fix = np.zeros(len(idx), dtype=np.int64)       # deferred replay: longest pending replay of the block
        second = np.zeros(len(idx), dtype=bool)        # second group of a trip (its start state sits in set B)
        while True:
            can = act & (n + G <= T)
            if not can.any():
                break
            co[can] += 6 * G + 2
            trip = alive & (esc > n[:, None]) & (esc <= (n + G)[:, None]) & can[:, None]   # lanes that really escape
            anyt = trip.any(1)
            mx = np.where(trip, esc, 0).max(1)
            if args.replay == "spot":
                co[anyt] += 9 * (mx[anyt] - n[anyt])
            else:
                co[anyt] += np.where(second[anyt], 5, 1)
                fix[anyt] = np.maximum(fix[anyt], mx[anyt] - n[anyt])
            second = np.where(can, ~second, second)
            alive &= ~trip
            alive &= ~((x <= (n + G)[:, None]) & can[:, None])                               # retired lanes leave too
            n = np.where(can, n + G, n)
            act = alive.any(1)
        while True:                                     # single-step remainder
            can = act & (n < T)
            if not can.any():
                break
            co[can] += 8
            n = np.where(can, n + 1, n)
            alive &= ~((x <= n[:, None]) & can[:, None])
            act = alive.any(1)
        co += np.where(fix > 0, 9 * fix + 4, 3)         # the fix-up: 3 instructions to find that nobody is pending
        cost[idx] += co
    return cost


strict = grouped_cost(np.full(B.shape, 10 ** 9))
fixed = args.fixed * len(B)
print(f"strict: loops {strict.sum()/1e6:.1f} M + fixed {fixed/1e6:.1f} M = {(strict.sum()+fixed)/1e6:.1f} M VALU per launch")
inset = (B == 0).all(1)
mid = ~inset & (last > E)
print(f"  all-in-set blocks {int(inset.sum())}: {strict[inset].sum()/1e6:.1f} M; boundary blocks {int(mid.sum())}: "
      f"{strict[mid].sum()/1e6:.1f} M (lock-step floor {last[mid].sum()*6.125/1e6:.1f} M); blocks done within {E} steps "
      f"{int((last <= E).sum())}: {strict[last <= E].sum()/1e6:.1f} M")

cc, ex = o.view_cycle(sr, si, rr, ri, N, N, args.mrd, first=E, check=args.check, window_cap=args.window_cap)
assert np.array_equal(cc, c)
never = c == 0
print(f"cycle test: {100*(ex[never] < T).mean():.1f} % of the {int(never.sum())} never-escaping pixels retire early "
      f"(mean {ex[never].mean():.0f} of {T} steps)")
X = blocks(ex).astype(np.int64)
lastc = X.max(1)
cyc = grouped_cost(np.where(B == 0, X, 10 ** 9))
cyc_checks = 2 * np.maximum(lastc - E, 0) // args.check   # two compares per --check steps in the grouped loops
print(f"  wave-steps {last.sum()/1e6:.2f} M -> {lastc.sum()/1e6:.2f} M; VALU {(strict.sum()+fixed)/1e6:.1f} M -> "
      f"{(cyc.sum()+cyc_checks.sum()+fixed)/1e6:.1f} M per launch")

"""CPU model for VERDICT r4 item 5 (cfg3, the deep-zoom config): split the 8x8 blocks by a probe -- blocks in which every
pixel stays inside run on `group` as they do (lane activity 1), only the blocks with an escaping pixel go to a lane-refill
kernel (`refill`: mbk_persist.h) -- on the exact counts of the full 8192^2 view.  Wave-steps (64 lanes x one step of the
loop) of: the product (one wave per block, lock-step), the split with an ORACLE classifier (knows which blocks are
all-alive), the split with the probes a launch can afford (centre / centre + corners after P steps), and what the refill
side must achieve for the whole to gain 4 %.  Overheads from the chip: refill runs 8-step groups (6.25 issue slots per
step against 6.125), measured lane activity 0.886 and VALU-busy 0.90 against 0.96 for `group` (profiles/r04
cfg3 PMC files, NOTES round 3-4).
    python scripts/split_refill_model.py [--n 8192]"""
import argparse
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle.oracle import COracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=8192)
ap.add_argument("--mrd", type=int, default=10000)
args = ap.parse_args()
o = COracle()
N, mrd, T = args.n, args.mrd, args.mrd - 1
view = (-0.743648, 0.131820, 1e-5, 1e-5)
t0 = time.time()
c = o.view_avx512(*view, N, N, mrd)[0] if o.have_avx512() else o.view(*view, N, N, mrd, want_bytes=False)[0]
print(f"cfg3 {N}^2 mrd {mrd}: oracle {time.time() - t0:.0f} s")
nb = N // 8
S = np.where(c == 0, T, c).astype(np.int64).reshape(nb, 8, nb, 8).transpose(0, 2, 1, 3).reshape(-1, 64)   # steps per lane
last = S.max(1)
lane_sum = S.sum(1)
ideal = lane_sum.sum() / 64.0
single = last.sum()
alive_all = (S.min(1) == T)
print(f"blocks {len(S)}; wave-steps: ideal {ideal / 1e6:.1f} M, one wave per block {single / 1e6:.1f} M -> lane activity {ideal / single:.4f}")
print(f"all-alive blocks: {alive_all.sum()} ({alive_all.mean():.3f}) holding {last[alive_all].sum() / single:.3f} of the wave-steps at activity 1")
rest = ~alive_all
act_rest = lane_sum[rest].sum() / 64.0 / last[rest].sum()
print(f"blocks with an escaping pixel: {rest.sum()} holding {last[rest].sum() / single:.3f} of the wave-steps at lane activity {act_rest:.4f}")
# within the rest: how is the idle lane-time distributed?
idle = last[rest] * 64 - lane_sum[rest]
order = np.argsort(-idle)
cum = np.cumsum(idle[order]) / idle.sum()
for q in (0.5, 0.8, 0.9):
    k = int(np.searchsorted(cum, q)) + 1
    print(f"   {q:.0%} of the idle lane-steps sit in {k} blocks ({k / len(S):.3%} of all)")
full_rest = (last[rest] == T)
print(f"   of those blocks {full_rest.sum()} still run all {T} steps (a pixel of the set in them): {last[rest][full_rest].sum() / single:.3f} of the wave-steps, activity "
      f"{lane_sum[rest][full_rest].sum() / 64.0 / last[rest][full_rest].sum():.4f}")


def total_with_refill(mask, act_refill, slots_refill=6.25, busy_ratio=0.90 / 0.96):
    """issue time (in `group` wave-steps) if the blocks in `mask` ran on a refill kernel with that lane activity"""
    g = last[~mask].sum()
    r = lane_sum[mask].sum() / 64.0 / act_refill * (slots_refill / 6.125) / busy_ratio
    return g + r


print("split with an ORACLE classifier (every block with an escaping pixel -> refill):")
for act in (0.886, 0.92, 0.95, 1.0):
    tt = total_with_refill(rest, act)
    t2 = total_with_refill(rest, act, busy_ratio=1.0)
    print(f"   refill lane activity {act:.3f}: total {tt / single:.4f} of the product's wave-steps with refill's measured VALU-busy 0.90 (x{single / tt:.3f}); {t2 / single:.4f} if it were as busy as group (x{single / t2:.3f})")
need = None
for act in np.arange(0.80, 1.0001, 0.002):
    if single / total_with_refill(rest, act) >= 1.04:
        need = act
        break
print(f"   for +4 % net the refill side needs lane activity >= {need if need is None else round(float(need), 3)} at its measured busy / slots")
need2 = None
for act in np.arange(0.80, 1.0001, 0.002):
    if single / total_with_refill(rest, act, busy_ratio=1.0) >= 1.04:
        need2 = act
        break
print(f"   ... and >= {need2 if need2 is None else round(float(need2), 3)} if it were as busy as group")
# affordable probes: the centre (and corners) after P steps -- what do they know about "all alive"?
C = c.reshape(nb, 8, nb, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
pts = {"centre": [36], "centre + corners": [36, 0, 7, 56, 63], "centre + corners + edge mids": [36, 0, 7, 56, 63, 3, 31, 32, 60]}
for P in (32, 512, 2048):
    for label, idx in pts.items():
        a = ((C[:, idx] == 0) | (C[:, idx] >= P)).all(1)          # the probe calls the block "alive"
        # blocks called alive stay on group at their real (lock-step) cost; the others go to refill
        tt = total_with_refill(~a, 0.886)
        cost = len(S) / 64.0 * len(idx) * P                        # probe wave-steps (one lane per block and point)
        print(f"   probe {label:28s} P = {P:5d}: calls {a.mean():.3f} of the blocks alive ({(a & rest).sum()} of them wrongly); total with refill(0.886) "
              f"{tt / single:.4f} + probe {cost / single:.4f} -> x{single / (tt + cost):.3f}")

# ---- what can a refill policy reach on the blocks with an escaping pixel?  Event simulation of persistent waves on the exact
# lifetimes: a wave holds 64 pixels, runs until at most `livemin` are left (exits are taken at 8-step group boundaries), then
# retires the finished ones and refills every free lane from its stream of blocks at a cost of `c` wave-steps per event (the
# slow path: retire stores, ranks, coordinates, queue pop -- mbk_persist.h says a refill must cost < ~12 steps to pay; the
# measured kernel: ~40 VALU + a scattered store per event, plus the wait for the pop).  8 192 waves share the stream; 96 of
# them, evenly spaced, are simulated and scaled.
rest_idx = np.nonzero(rest)[0]
life = S[rest_idx]                                   # [block, 64] steps per pixel, in the order the blocks would be popped
NW, SIM = 8192, 96
per_wave = len(rest_idx) // NW


def simulate(livemin, c, group=8):
    """-> (wave-steps, refill events) of the refill policy on ALL the blocks with an escaping pixel: the sampled waves' ratio
    to their own lock-step cost, applied to the whole (the streams differ: a wave's blocks are neighbours)"""
    steps = events = lock_sim = 0
    for w in np.linspace(0, NW - 1, SIM).astype(int):
        blocks = life[w * per_wave:(w + 1) * per_wave]
        lock_sim += int(blocks.max(1).sum())
        stream = blocks.reshape(-1)
        pos = 64
        rem = stream[:64].astype(np.int64).copy()
        live = np.ones(64, bool)
        while True:
            nlive = int(live.sum())
            if nlive == 0:
                break
            r = np.sort(rem[live])
            more = pos < len(stream)
            # run until only `livemin` lanes are left (or, with nothing to refill from, until the last one is done)
            k = nlive - livemin - 1 if (more and nlive > livemin) else nlive - 1
            dt = int(-(-r[max(k, 0)] // group) * group)
            steps += dt
            rem -= dt
            live &= rem > 0
            if more:
                free = np.nonzero(~live)[0]
                take = min(len(free), len(stream) - pos)
                if take:
                    rem[free[:take]] = stream[pos:pos + take]
                    live[free[:take]] = True
                    pos += take
                    events += 1
        # (the simulated wave's share of the tail -- lanes draining with nothing left to pop -- is in `steps`)
    scale = float(last[rest].sum()) / lock_sim
    return steps * scale, events * scale


lock = last[rest].sum()
ideal_rest = lane_sum[rest].sum() / 64.0
print(f"refill policies on the {rest.sum()} blocks with an escaping pixel (lock-step {lock / 1e6:.1f} M wave-steps at activity {ideal_rest / lock:.3f}; ideal {ideal_rest / 1e6:.1f} M):")
print("   livemin  wave-steps   events      activity | total (group for the all-alive blocks + this) relative to the product, refill at 6.25 / 6.125 slots, for an event cost of")
print("                                               |    c = 6      c = 12      c = 24      c = 48 wave-steps        ... and with refill's measured VALU-busy 0.90 / 0.96 on top (c = 12)")
g_all = last[alive_all].sum()
for livemin in (32, 40, 48, 56, 60):
    st, ev = simulate(livemin, 0)
    cells = []
    for cst in (6, 12, 24, 48):
        tot = g_all + (st + ev * cst) * (6.25 / 6.125)
        cells.append(f"x{single / tot:.3f}")
    tot_busy = g_all + (st + ev * 12) * (6.25 / 6.125) / (0.90 / 0.96)
    print(f"   {livemin:5d}   {st / 1e6:8.1f} M  {ev / 1e6:6.2f} M   {ideal_rest / st:8.3f} |  " + "     ".join(cells) + f"          x{single / tot_busy:.3f}")

# ---- the library's default (cycle test): the all-alive side keeps the cycle test (kernel `group`), the refill side has none --
# its never-escaping pixels (few: they sit in blocks with an escaping pixel) run all mrd - 1 steps, each in a lane of its own
t0 = time.time()
ex = o.view_cycle(*view, N, N, mrd, first=8, check=8)[1]
E = np.where(c == 0, ex, c).astype(np.int64).reshape(nb, 8, nb, 8).transpose(0, 2, 1, 3).reshape(-1, 64)   # executed steps per lane
last_c = E.max(1)
single_c = last_c.sum()
print(f"cycle test (oracle.view_cycle, {time.time() - t0:.0f} s): one wave per block {single_c / 1e6:.1f} M wave-steps (lane activity {E.sum() / 64.0 / single_c:.4f}); "
      f"the all-alive blocks hold {last_c[alive_all].sum() / single_c:.3f} of them at activity {E[alive_all].sum() / 64.0 / last_c[alive_all].sum():.3f}")
for livemin in (40, 48):
    st, ev = simulate(livemin, 0)          # the refill side runs the strict steps of its pixels
    for cst in (12, 24):
        tot = last_c[alive_all].sum() + (st + ev * cst) * (6.25 / 6.125)
        print(f"   hybrid, refill without cycle test, livemin {livemin}, event cost {cst}: {tot / 1e6:.1f} M = x{single_c / tot:.3f} of the product's cycle-test launch "
              f"(refill side {(st + ev * cst) / 1e6:.1f} M against {last_c[rest].sum() / 1e6:.1f} M now)")

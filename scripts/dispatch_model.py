"""What does the dispatch order cost?  (CPU only; companion of scripts/cycle_model.py.)

    python scripts/dispatch_model.py [--mrd 1000] [--size 4096]

Event simulation of one cfg2 launch of the one-wave-per-block kernels: 1024 SIMDs, 8 wave slots each, processor
sharing inside a SIMD (a lone wave can take at most half of its issue rate), a freed slot takes the next workgroup
of the order list.  Block durations are instruction counts from the tile's exact counts (steps of the slowest
lane x 6.2 + 40; with the cycle test: oracle.view_cycle).  Output: makespan / (total work / 1024) per order.
The model reproduces the measured idle share of both kernels (strict: 1.040 vs 93.4 % VALU-busy + classify;
cycle test: 1.088 vs 88.8 %), and says where the rest of it is: not in the order of the in-set blocks but in the
boundary blocks the one-pixel probe files under "light" and therefore dispatches last."""
import argparse
import heapq
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle.oracle import COracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mrd", type=int, default=1000)
ap.add_argument("--size", type=int, default=4096)
args = ap.parse_args()
o = COracle()
N, T, E = args.size, args.mrd - 1, 8
view = (-2.0, -1.5, 3.0, 3.0)
c = o.view_avx512(*view, N, N, args.mrd)[0] if o.have_avx512() else o.view(*view, N, N, args.mrd, want_bytes=False)[0]
nb = N // 8


def blocks(x):
    return x.reshape(nb, 8, nb, 8).transpose(0, 2, 1, 3).reshape(-1, 64)


B = blocks(c)
center = c[4::8, 4::8].reshape(-1)
heavy = (center == 0) | (center >= 32)            # classify_blocks_kernel: centre pixel alive after 32 steps
_, ex = o.view_cycle(*view, N, N, args.mrd, first=E, check=16)
steps = {"strict": np.minimum(np.where(B == 0, 10 ** 9, B).max(1), T), "cycle test": blocks(np.where(c == 0, ex, c)).max(1)}
NS, SL, CAP = 1024, 8, 0.5


def simulate(dur, order):
    n, nxt = len(order), 0
    v, k, tlast, stamp = [0.0] * NS, [0] * NS, [0.0] * NS, [0] * NS
    jobs = [[] for _ in range(NS)]                # per SIMD: heap of virtual finish times
    ev = []

    def rate(kk):
        return min(1.0 / kk, CAP) if kk else 0.0

    def push(s, now):
        stamp[s] += 1
        if jobs[s]:
            heapq.heappush(ev, (now + (jobs[s][0] - v[s]) / rate(k[s]), s, stamp[s]))

    for _ in range(SL):
        for s in range(NS):
            if nxt < n:
                heapq.heappush(jobs[s], v[s] + dur[order[nxt]])
                k[s] += 1
                nxt += 1
    for s in range(NS):
        push(s, 0.0)
    now = 0.0
    while ev:
        t, s, st = heapq.heappop(ev)
        if st != stamp[s]:
            continue
        v[s] += (t - tlast[s]) * rate(k[s])
        tlast[s] = now = t
        heapq.heappop(jobs[s])
        k[s] -= 1
        if nxt < n:
            heapq.heappush(jobs[s], v[s] + dur[order[nxt]])
            k[s] += 1
            nxt += 1
        push(s, t)
    return now / (dur.sum() / NS)


idx = np.arange(len(B))
for name, st in steps.items():
    dur = (st * 6.2 + 40).astype(float)
    mid = (~heavy) & (center >= 6)
    orders = {
        "image order": idx,
        "heavy first (what the kernels do)": np.concatenate([idx[heavy], idx[~heavy][::-1]]),
        "heavy, then probe count >= 6, then the rest": np.concatenate([idx[heavy], idx[mid], idx[(~heavy) & (~mid)][::-1]]),
        "longest first (needs the answer)": np.argsort(-dur, kind="stable"),
    }
    for k_, od in orders.items():
        print(f"{name:10s} {k_:46s} makespan / ideal {simulate(dur, od):.3f}")

"""Event model of one launch of the one-wave-per-block kernels with what round 4 measured about the chip (CPU only).

    python scripts/fifo_model.py [--workload cfg2|chunk_l1] [--orders ...]

Round 3's model (dispatch_model.py) let the waves of a SIMD share it equally.  profiles/r04/valu_issue.txt shows that
the SIMD arbiter serves its OLDEST wave first: of 8 resident waves running the same loop, one finishes after the
other, each at (nearly) the single-wave rate.  This model therefore has, per SIMD, an age-ordered queue in which the
oldest wave with vector work gets `lead` of the issue slots and the next one the rest, and around it
  * 1024 SIMDs x 8 slots, workgroups handed out in list order by ONE dispatcher at `disp_ns` per workgroup (0.28 ns:
    the all-exterior tile takes 73 us for 262 144 workgroups at 8 and at 6 waves per SIMD alike, profiles/r04);
  * every wave holds its slot for `start_ns` before its first vector instruction (launch, kernel arguments, list entry)
    and `end_ns` after its last (stores drain): 1.0 us in all for a light block alone on the chip (same sweep:
    wave_limit 2 and 3 are slot-bound at 0.98-1.10 us per workgroup);
  * vector work per block from the tile's exact counts (steps of the slowest lane x 6.2 + 40 instructions) at 4.06
    cycles per instruction and the clock the probe measured under load (2.34 GHz).
It answers: how long does the launch take for a given order of the workgroups, and where does the time beyond
(total vector work / 1024 SIMDs) go -- the dispatch-bound light phase, the drain of the last waves, or slots.
"""
import argparse
import heapq
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle.oracle import COracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2", choices=["cfg2", "chunk_l1"])
ap.add_argument("--mrd", type=int, default=1000)
ap.add_argument("--clock", type=float, default=2.34)
ap.add_argument("--cpi", type=float, default=4.06)
ap.add_argument("--lead", type=float, default=0.93, help="share of the issue slots the oldest wave takes (4.06 / 4.38)")
ap.add_argument("--disp-ns", type=float, default=0.28)
ap.add_argument("--start-ns", type=float, default=700.0)
ap.add_argument("--end-ns", type=float, default=300.0)
ap.add_argument("--slots", type=int, default=8)
ap.add_argument("--light-k", type=int, default=0, help="also model light workgroups that take K list entries each")
ap.add_argument("--cycle", action="store_true", help="the library's default: the cycle test retires periodic orbits (oracle.view_cycle: executed steps per pixel)")
ap.add_argument("--study", default="orders", choices=["orders", "tail"], help="tail: which class should be dispatched first (end of round 4)")
ap.add_argument("--sub", type=int, default=8, help="simulate 1024 / SUB SIMDs with every SUB-th workgroup of the order (dispatcher slowed by SUB)")
args = ap.parse_args()

VIEWS = {"cfg2": (-2.0, -1.5, 3.0, 3.0), "chunk_l1": (-2.0, -2.0, 4.0, 4.0)}
o = COracle()
N, T = 4096, args.mrd - 1
view = VIEWS[args.workload]
c = o.view_avx512(*view, N, N, args.mrd)[0] if o.have_avx512() else o.view(*view, N, N, args.mrd, want_bytes=False)[0]
nb = N // 8
B = c.reshape(nb, 8, nb, 8).transpose(0, 2, 1, 3).reshape(-1, 64)
center = c[4::8, 4::8].reshape(-1)
heavy = (center == 0) | (center >= 32)            # classify_blocks_kernel: centre pixel alive after 32 steps
last = np.minimum(np.where(B == 0, 10 ** 9, B).max(1), T)
if args.cycle:
    # a lane leaves at its escape step or when the cycle test retires it; the wave runs until its last lane has left
    ex = o.view_cycle(*view, N, N, args.mrd, first=8, check=8)[1]
    last = np.minimum(ex.reshape(nb, 8, nb, 8).transpose(0, 2, 1, 3).reshape(-1, 64).max(1), T)
instr = last * 6.2 + 40.0
NS = 1024 // args.sub
NS_PER_INSTR = args.cpi / args.clock              # ns of one SIMD per wave-instruction


def simulate(order_instr, slots=args.slots, start_ns=args.start_ns, end_ns=args.end_ns, disp_ns=args.disp_ns, lead=args.lead):
    """order_instr: vector instructions of the workgroups in dispatch order.  Returns (makespan ns, time the last workgroup
    was dispatched, vector-busy share)."""
    n = len(order_instr)
    # per SIMD: list of waves [ready_time, remaining_instr] in age order; a wave occupies a slot from dispatch to exit
    q = [[] for _ in range(NS)]
    tlast = [0.0] * NS
    free = [slots] * NS
    nfree = [slots * NS]
    stamp = [0] * NS
    ev = []                         # (time, kind, simd, stamp): kind 0 = a wave of the SIMD finishes its work / becomes ready
    exits = []                      # (time, simd): slot frees
    busy = 0.0

    def advance(s, now):
        """run SIMD s from tlast[s] to now: the oldest ready wave at `lead`, the next ready one at 1 - lead"""
        nonlocal busy
        t = tlast[s]
        while t < now - 1e-9:
            ready = [w for w in q[s] if w[0] <= t + 1e-9 and w[1] > 0]
            nxt_ready = min([w[0] for w in q[s] if w[0] > t + 1e-9 and w[1] > 0], default=now)
            horizon = min(now, nxt_ready)
            if not ready:
                t = horizon
                continue
            a = ready[0]
            b = ready[1] if len(ready) > 1 else None
            ra = lead if b is not None else lead      # alone: the single-wave rate
            rb = 1.0 - lead
            ta = a[1] * NS_PER_INSTR / ra
            tb = b[1] * NS_PER_INSTR / rb if b is not None else 1e30
            dt = min(horizon - t, ta, tb)
            a[1] -= dt * ra / NS_PER_INSTR
            busy += dt * ra
            if b is not None:
                b[1] -= dt * rb / NS_PER_INSTR
                busy += dt * rb
            t += dt
            for w in (a, b):
                if w is not None and w[1] <= 1e-6 and w[1] > -1:
                    w[1] = -2.0                  # done: its slot frees end_ns later
                    heapq.heappush(exits, (t + end_ns, s))
        q[s] = [w for w in q[s] if w[1] > 0]
        tlast[s] = now

    def next_event(s, now):
        """time at which something changes on SIMD s (a wave finishes or becomes ready)"""
        ready = [w for w in q[s] if w[0] <= now + 1e-9 and w[1] > 0]
        cand = [w[0] for w in q[s] if w[0] > now + 1e-9]
        if ready:
            a = ready[0]
            cand.append(now + a[1] * NS_PER_INSTR / lead)
            if len(ready) > 1:
                cand.append(now + ready[1][1] * NS_PER_INSTR / (1.0 - lead))
        return min(cand) if cand else None

    now, nxt, tdisp, t_last_disp = 0.0, 0, 0.0, 0.0
    rr = 0
    while nxt < n or exits or any(q[s] for s in range(NS)):
        # dispatch as long as the dispatcher is free and some SIMD has a slot
        progressed = False
        while nxt < n and tdisp <= now + 1e-9:
            # round-robin over SIMDs with a free slot
            if nfree[0] == 0:
                break
            while free[rr] == 0:
                rr = (rr + 1) % NS
            s = rr
            rr = (rr + 1) % NS
            advance(s, now)
            free[s] -= 1
            nfree[0] -= 1
            q[s].append([now + start_ns, float(order_instr[nxt])])
            stamp[s] += 1
            te = next_event(s, now)
            if te is not None:
                heapq.heappush(ev, (te, s, stamp[s]))
            nxt += 1
            tdisp = max(tdisp, now) + disp_ns
            t_last_disp = now
            progressed = True
            if tdisp > now + 1e-9:
                break
        # next time something happens
        cands = []
        if nxt < n and tdisp > now + 1e-9 and nfree[0] > 0:
            cands.append(tdisp)
        while ev and ev[0][2] != stamp[ev[0][1]]:
            heapq.heappop(ev)
        if ev:
            cands.append(ev[0][0])
        if exits:
            cands.append(exits[0][0])
        if not cands:
            break
        now = max(now, min(cands))
        while exits and exits[0][0] <= now + 1e-9:
            _, s = heapq.heappop(exits)
            free[s] += 1
            nfree[0] += 1
        while ev and ev[0][0] <= now + 1e-9:
            _, s, st = heapq.heappop(ev)
            if st != stamp[s]:
                continue
            advance(s, now)
            stamp[s] += 1
            te = next_event(s, now)
            if te is not None:
                heapq.heappush(ev, (te, s, stamp[s]))
    return now, t_last_disp, busy / (now * NS)


def report(name, order_instr):
    order_instr = np.asarray(order_instr, dtype=float)[::args.sub]
    ms, tl, b = simulate(order_instr, disp_ns=args.disp_ns * args.sub)
    ideal = float(np.sum(order_instr)) * NS_PER_INSTR / NS
    print(f"{args.workload:9s} {name:58s} {ms / 1e3:8.1f} us   vector work / 1024 = {ideal / 1e3:6.1f} us   last dispatch at {tl / 1e3:6.1f} us   "
          f"x{ms / ideal:.3f}", flush=True)


idx = np.arange(len(B))
h, l = idx[heavy], idx[~heavy][::-1]
if args.study == "tail":
    # The units order as built (H = centre alive at 32 steps, M = centre gone at step 4..31, V = gone within 3 steps, eight to a
    # workgroup) against orders that send the straggler-prone blocks out earlier.  Motivation (profiles/r04/units_skew_cycle.txt):
    # with the cycle test every XCD's last wave ends 15-25 us (6-8 % of the launch) after its dispatcher ran dry -- boundary
    # blocks of class M, dispatched behind all of H, that run (nearly) all mrd - 1 steps, while the interior (H) retires early.
    vl = (~heavy) & (center >= 1) & (center <= 3)
    mid = (~heavy) & ~vl

    def runs_of(sel, k=8):
        li = np.where(last[idx[sel]] <= 4, 18.5, instr[idx[sel]] + 20.0)
        pad = (-len(li)) % k
        return np.concatenate([li, np.zeros(pad)]).reshape(-1, k).sum(1) + 40.0

    vruns = runs_of(vl)
    report("units as built: H, M, V runs", np.concatenate([instr[h], instr[idx[mid]], vruns]))
    report("M first, then H, then V runs", np.concatenate([instr[idx[mid]], instr[h], vruns]))
    # a richer probe: centre + four corners of the block, 32 steps; alive anywhere -> long
    probes = [c[4::8, 4::8], c[0::8, 0::8], c[0::8, 7::8], c[7::8, 0::8], c[7::8, 7::8]]
    alive5 = np.zeros(len(B), bool)
    for pr in probes:
        pr = pr.reshape(-1)
        alive5 |= (pr == 0) | (pr >= 32)
    m_long = mid & alive5
    print(f"classes: H {int(heavy.sum())}, M {int(mid.sum())} of which a corner is alive at 32 steps {int(m_long.sum())}, V blocks {int(vl.sum())} "
          f"(V blocks with a live corner: {int((vl & alive5).sum())})")
    report("H, M with a live corner, other M, V runs", np.concatenate([instr[h], instr[idx[m_long]], instr[idx[mid & ~m_long]], vruns]))
    report("M with a live corner, H, other M, V runs", np.concatenate([instr[idx[m_long]], instr[h], instr[idx[mid & ~m_long]], vruns]))
    # the bound: every single-block workgroup in order of its true cost (not realisable: needs the answer)
    singles = np.concatenate([instr[h], instr[idx[mid]]])
    report("all single blocks longest first (oracle order), V runs", np.concatenate([np.sort(singles)[::-1], vruns]))
    # M by the centre probe's escape step, latest first (what classify knows already)
    order_m = idx[mid][np.argsort(-center[idx[mid]], kind="stable")]
    report("H, M by centre escape step (latest first), V runs", np.concatenate([instr[h], instr[order_m], vruns]))
    report("M by centre escape step (latest first), H, V runs", np.concatenate([instr[order_m], instr[h], vruns]))
    # What the probe already knows at no extra cost: how fast the centre pixel's orbit is settling after its 32 steps,
    # delta = min over p in {1..6, 8} of |z_32 - z_(32-p)|^2.  With the cycle test an interior block whose orbit has settled
    # retires within a few checks, one near the boundary runs (nearly) all steps: H blocks with a large delta first.
    re = np.linspace(view[0], view[0] + view[2], N)[4::8]
    im = np.linspace(view[1], view[1] + view[3], N)[4::8]
    CR, CI = np.meshgrid(re, im)
    pcr, pci = CR.reshape(-1)[h], CI.reshape(-1)[h]
    zr, zi, hist = pcr.copy(), pci.copy(), {}
    with np.errstate(all="ignore"):
        for n in range(1, 33):
            zr, zi = zr * zr - zi * zi + pcr, 2 * zr * zi + pci
            if n >= 24:
                hist[n] = (zr.copy(), zi.copy())
        delta = np.full(len(h), np.inf)
        for p_ in (1, 2, 3, 4, 5, 6, 8):
            delta = np.minimum(delta, (hist[32][0] - hist[32 - p_][0]) ** 2 + (hist[32][1] - hist[32 - p_][1]) ** 2)
    delta = np.where(np.isfinite(delta), delta, 1e30)
    for thr in (1e-9, 1e-6, 1e-4):
        slow = delta > thr
        report(f"H unsettled (delta > {thr:g}: {int(slow.sum())}) first, settled H, M, V runs",
               np.concatenate([instr[h][slow], instr[h][~slow], instr[idx[mid]], vruns]))
    b3 = np.digitize(np.log10(delta + 1e-300), [-9.0, -6.0, -4.0])      # 0 settled .. 3 far from it
    report("H in four delta classes, least settled first, M, V runs",
           np.concatenate([instr[h][b3 == 3], instr[h][b3 == 2], instr[h][b3 == 1], instr[h][b3 == 0], instr[idx[mid]], vruns]))
    report("H sorted by delta (largest first), M, V runs", np.concatenate([instr[h][np.argsort(-delta, kind="stable")], instr[idx[mid]], vruns]))
    sys.exit(0)
inset = last[h] >= T
report("image order", instr)
report("heavy first (what the kernels do)", np.concatenate([instr[h], instr[l]]))
report("heavy sorted longest first, then light", np.concatenate([np.sort(instr[h])[::-1], instr[l]]))
report("in-set first, other heavy longest first, then light", np.concatenate([instr[h][inset], np.sort(instr[h][~inset])[::-1], instr[l]]))
report("heavy only (scan pass 2 without its light blocks)", instr[h])
report("heavy only, longest first", np.sort(instr[h])[::-1])
if args.light_k:
    k = args.light_k
    li = instr[l] - 40.0 + 15.0                  # a light block inside a K-entry workgroup: no per-workgroup overhead
    pad = (-len(li)) % k
    grouped = np.concatenate([li, np.zeros(pad)]).reshape(-1, k).sum(1) + 40.0
    report(f"heavy first, light blocks {k} per workgroup", np.concatenate([instr[h], grouped]))
    report(f"heavy longest first, light blocks {k} per workgroup", np.concatenate([np.sort(instr[h])[::-1], grouped]))

# ---- "units" design (round 4 study): heavy blocks and boundary blocks one per workgroup, VERY light blocks (centre pixel gone
# within 3 steps) in runs of K per workgroup through the light path (18.5 instructions per block that four steps finish; a
# block of the run that they do not finish is finished in place at the price of a whole block)
vl = (~heavy) & (center >= 1) & (center <= 3)
mid = (~heavy) & ~vl
for k in (4, 8, 16):
    li = np.where(last[idx[vl]] <= 4, 18.5, instr[idx[vl]] + 20.0)
    pad = (-len(li)) % k
    runs = np.concatenate([li, np.zeros(pad)]).reshape(-1, k).sum(1) + 40.0
    report(f"units: heavy, boundary singles, very light in runs of {k}", np.concatenate([instr[h], instr[idx[mid]], runs]))
    report(f"units: heavy, runs of {k} interleaved with boundary singles",
           np.concatenate([instr[h], np.random.RandomState(1).permutation(np.concatenate([instr[idx[mid]], runs]))]))
print("classes: heavy", int(heavy.sum()), "boundary (centre gone at step 4..31)", int(mid.sum()), "very light", int(vl.sum()),
      "of which finished by 4 steps", int((last[idx[vl]] <= 4).sum()))
# round 3's measured experiment, for calibration: three classes of SINGLE blocks (heavy / probe count >= 6 / rest): measured +0.7 % slower than two classes
mid6 = (~heavy) & (center >= 6)
report("r3 experiment: heavy, probe count >= 6 singles, light singles", np.concatenate([instr[h], instr[idx[mid6]], instr[idx[(~heavy) & ~mid6]][::-1]]))

#!/usr/bin/env python3
"""scale_emulate.py -- predict bench.py's 1/2/4/8-GPU curve from ONE GPU.

    python scripts/scale_emulate.py [--out profiles/r04/scale_prediction.json] [--jobs queue,cfg3_grid2,cfg3_grid4,cfg3_bands]
                                    [--worlds 1,2,4,8]

The N > 1 modes of bench.py have no data-path collective: the ranks pull tickets from one shared cursor and every
GPU works through what it drew, alone.  A run on N GPUs is therefore N independent GPU timelines plus a schedule, and
both can be had on one GPU:

  1. measure every unit of the job alone (a tile of `--shard queue`, a row band of `--shard bands`);
  2. replay the cursor for an emulated world of N ranks with those durations (event simulation: a rank holds `streams`
     tickets in flight and pulls the next one the moment a launch of its own completes -- the policy of bench.py's
     run_steps, including the longest-first tile order and the guided chunks of the bands mode, through the same
     `next_guided` arithmetic);
  3. execute each emulated rank's picks on the real GPU, back to back, with the same streams in flight, and time them;
     the predicted job time is the slowest rank's, the predicted `value` the job's pixel-iterations over that time.

What this captures: load imbalance and the tail, launch-size effects (a band of a deep zoom cannot be shorter than
its slowest block), per-launch overheads, the kernels' real durations when they follow each other.  What it does not:
N Python processes contending for the cursor's fcntl lock and for host cores (a pull costs ~10 us against launches of
30 us .. 14 ms), the ranks' start skew after the barrier (< 1 ms), and box-to-box clock differences (each GPU of a
node has its own governor; the pool's boxes differ by 3-4 %).  `efficiency` = predicted value(N) / (N x value(1)),
value(1) being the same replay with one rank -- the quantity bench.py's `efficiency_same_job` measures on hardware.
"""
from __future__ import annotations

import argparse
import heapq
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (WORKLOADS, DEFAULT_STEPS)


class LocalCursor:
    """sharding.SharedCursor's arithmetic without the file (one process plays every rank)."""

    def __init__(self):
        self.v = 0

    def next(self):
        v = self.v
        self.v += 1
        return v

    def next_guided(self, limit, divisor, period=0):
        v = self.v
        if v >= limit:
            return limit, 0
        k = max(1, (limit - v) // max(1, divisor))
        if period > 0:
            k = min(k, period - v % period)
        k = min(k, limit - v)
        self.v = v + k
        return v, k


def simulate(world, nstreams, nsteps, nunits, unit_ms, guided):
    """Event simulation of bench.py's run_steps for `world` ranks.  Returns per-rank lists of picks; a pick is
    (first ticket, count).  unit_ms[u] = measured duration of unit u alone; a GPU runs its launches one after the other
    (a launch fills the chip), a slot is free again when its launch has completed."""
    cur = LocalCursor()
    limit = nsteps * nunits
    picks = [[] for _ in range(world)]
    gpu_free = [0.0] * world          # when the GPU of rank r has finished everything launched so far
    events = []                       # (time a slot of rank r frees, tiebreak, r)
    seq = 0
    for s in range(nstreams):         # every rank fills its slots at the start, ranks interleaved as they race
        for r in range(world):
            heapq.heappush(events, (1e-6 * seq, seq, r))
            seq += 1
    while events:
        t, _, r = heapq.heappop(events)
        if guided:
            first, k = cur.next_guided(limit, 2 * world, period=nunits)
            if k == 0:
                continue
        else:
            first, k = cur.next(), 1
            if first >= limit:
                continue
        picks[r].append((first, k))
        dur = sum(unit_ms[(first + j) % nunits] for j in range(k))
        gpu_free[r] = max(gpu_free[r], t) + dur
        heapq.heappush(events, (gpu_free[r], seq, r))
        seq += 1
    return picks, gpu_free


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r04", "scale_prediction.json"))
    ap.add_argument("--jobs", default="queue,queue_longest,cfg3_grid2,cfg3_grid4,cfg3_bands")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--max-steps", type=int, default=0, help="cap the emulated steps per job (0 = bench.py's defaults)")
    ap.add_argument("--kernel", default="default")
    args = ap.parse_args()
    worlds = [int(x) for x in args.worlds.split(",")]

    import torch
    from distributedmandelbrot_amd import MandelbrotDevice, View
    from distributedmandelbrot_amd.sharding import Band, make_bands

    dev = MandelbrotDevice(0)
    dev.set_option("cycle_detect", 0)        # the headline's setting: every iteration executed
    torch.cuda.set_device(0)
    results = {"device": dev.info(), "pci_bus_id": dev.pci_bus_id(), "kernel": args.kernel, "cycle_test": "off (as bench.py's value)",
               "method": __doc__.split("\n\n")[2].strip(), "not_modelled": "cursor lock / host contention between N processes, "
               "start skew after the barrier, clock differences between the GPUs of a node", "jobs": {}}

    def sync():
        torch.cuda.synchronize()

    for job in args.jobs.split(","):
        base_job = job[:-len("_longest")] if job.endswith("_longest") else job     # NAME_longest: --queue-order longest
        if base_job == "queue":
            wl, mode, grid = "cfg2", "queue", 8
        elif base_job.startswith("cfg3_grid"):
            wl, mode, grid = "cfg3", "queue", int(base_job[len("cfg3_grid"):])
        elif job == "cfg3_bands":
            wl, mode, grid = "cfg3", "bands", 0
        elif job == "cfg2_bands":
            wl, mode, grid = "cfg2", "bands", 0
        else:
            raise SystemExit(f"unknown job {job}")
        sr, si, rng, width, height, mrd, desc = bench.WORKLOADS[wl]
        npix = width * height
        nstreams = 4 if mode == "queue" else 3                      # bench.py's defaults
        d_steps = bench.DEFAULT_STEPS[wl][0]
        view = View(sr, si, rng, rng, width, height)
        qview = View(sr, si, rng, rng, width * max(grid, 1), height * max(grid, 1))
        bufs = [torch.empty(npix, dtype=torch.int32, device="cuda:0") for _ in range(nstreams)]
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
        ev = [torch.cuda.Event() for _ in range(nstreams)]
        busy = [False] * nstreams

        def launch_tile(i, u):
            tr, ti = u % grid, u // grid
            dev.launch_view(qview, mrd, window=(tr * width, ti * height, width, height), d_counts=bufs[i].data_ptr(),
                            stream=streams[i].cuda_stream, kernel=args.kernel)

        def launch_rows(i, row0, nrows):
            dev.launch_view(view, mrd, window=(0, row0, width, nrows), d_counts=bufs[0].data_ptr() + 4 * row0 * width,
                            stream=streams[i].cuda_stream, kernel=args.kernel)

        def timed_alone(fn, reps=2):
            best = None
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(streams[0])
                fn()
                e1.record(streams[0])
                sync()
                ms = e0.elapsed_time(e1)
                best = ms if best is None else min(best, ms)
            return best

        def ramp(fn, ms=150.0):
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < ms:
                fn()
                sync()

        # ---- the job's units, their work and their durations alone ----
        if mode == "queue":
            ntiles = grid * grid
            iters = {}
            ramp(lambda: launch_tile(0, ntiles // 2))
            unit_ms = [0.0] * ntiles
            for u in range(ntiles):
                unit_ms[u] = timed_alone(lambda: launch_tile(0, u))
                iters[u] = dev.reduce_counts(bufs[0].data_ptr(), npix, mrd, stream=streams[0].cuda_stream).pixel_iterations
            # bench.py --queue-order: image (default since round 5: the server's walk) | longest (job name ending in _longest)
            order = sorted(range(ntiles), key=lambda u: (-iters[u], u)) if job.endswith("_longest") else list(range(ntiles))
            unit_ms_in_order = [unit_ms[u] for u in order]
            iters_per_step = sum(iters.values())
            nunits = ntiles
            base_steps = max(2, d_steps * 4 // ntiles)
            scheme = f"{ntiles} tiles of {width}x{height} per step ({grid}x{grid} grid over the {wl} region), {'longest first' if job.endswith('_longest') else 'image order'}, {nstreams} in flight"
        else:
            ramp(lambda: launch_rows(0, 0, height))
            launch_rows(0, 0, height)
            sync()
            iters_per_step = dev.reduce_counts(bufs[0].data_ptr(), npix, mrd, stream=streams[0].cuda_stream).pixel_iterations
            base_steps = d_steps
            scheme = None   # depends on the world (band height = height / (16 N), >= 128 rows)

        job_rec = {"workload": f"{wl}: {desc}", "mode": mode, "pixel_iterations_per_step": iters_per_step, "worlds": {}}
        value1 = None
        for world in worlds:
            nsteps = base_steps * world
            if args.max_steps:
                nsteps = min(nsteps, args.max_steps * world)
            if mode == "bands":
                band_rows = max(8, (max(128, height // (16 * world)) // 8) * 8)
                bands = make_bands(height, band_rows)
                nunits = len(bands)
                ramp(lambda: launch_rows(0, 0, height), 100.0)
                unit_ms_in_order = [timed_alone(lambda b=b: launch_rows(0, b.row0, b.nrows), reps=1) for b in bands]
                scheme = f"one {width}x{height} image per step in {nunits} bands of {band_rows} rows, guided chunks, {nstreams} in flight"
            picks, sim_finish = simulate(world, nstreams, nsteps, nunits, unit_ms_in_order, guided=(mode == "bands"))
            rank_ms = []
            for r in range(world):
                ramp((lambda: launch_tile(0, order[0])) if mode == "queue" else (lambda: launch_rows(0, 0, height)), 120.0)
                turn = 0
                t0 = time.perf_counter()
                for first, k in picks[r]:
                    i = turn % nstreams
                    if busy[i]:
                        ev[i].synchronize()
                        busy[i] = False
                    if mode == "queue":
                        launch_tile(i, order[first % nunits])
                    else:
                        b0, b1 = bands[first % nunits], bands[first % nunits + k - 1]
                        launch_rows(i, b0.row0, b1.row0 + b1.nrows - b0.row0)
                    ev[i].record(streams[i])
                    busy[i] = True
                    turn += 1
                sync()
                busy = [False] * nstreams
                rank_ms.append((time.perf_counter() - t0) * 1e3)
            total_ms = max(rank_ms)
            value = iters_per_step * nsteps / total_ms / 1e6
            if world == 1:
                value1 = value
            rec = {"steps": nsteps, "scheme": scheme, "rank_ms": [round(x, 3) for x in rank_ms],
                   "simulated_finish_ms": [round(x, 3) for x in sim_finish], "launches_per_rank": [len(p) for p in picks[:world]],
                   "predicted_ms_per_step": total_ms / nsteps, "predicted_value_G_per_s": value}
            if value1:
                rec["predicted_speedup"] = value / value1
                rec["predicted_efficiency"] = value / (world * value1)
            job_rec["worlds"][str(world)] = rec
            print(f"{job:11s} N={world}: steps {nsteps:4d}  slowest rank {total_ms:9.2f} ms  (fastest {min(rank_ms):9.2f})  "
                  f"value {value:8.1f} G/s  speed-up {rec.get('predicted_speedup', 1.0):5.2f}  eff {rec.get('predicted_efficiency', 1.0):.3f}",
                  flush=True)
        results["jobs"][job] = job_rec
        del bufs
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(results, f, indent=1)
    print("wrote", args.out)
    dev.close()


if __name__ == "__main__":
    main()

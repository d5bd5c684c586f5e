#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Mandelbrot tile escape-time path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2] [--kernel default]

One "step" = one pass of the hot path over one synthetic tile already described in HBM terms: the
kernel generates its coordinates itself and writes int32 escape indices to a resident HBM buffer
(nothing crosses PCIe inside the timed region).  Workload at every N: BASELINE.json configs[1]
("cfg2"): 4096x4096 samples of the full set (centre -0.5+0i, span 3.0), max_iter (mrd) = 1000, fp64.
For N > 1 the driver launches one rank per GPU (torch.distributed.run); tiles are independent, so
every rank computes its own tile with no data-path collective ("weak" scaling); the only
communication is the barrier and the max-over-ranks of the elapsed time, on gloo (CPU tensors) -- there
is no RCCL anywhere (`--control nccl` exists to A/B that choice).  `--shard bands` is the strong-scaling
form BASELINE cfg3 asks for: ONE view per step, cut into >= 16 row bands per GPU (of >= 128 rows) which the ranks pull
from a cursor in shared memory (distributedmandelbrot_amd.sharding.SharedCursor; dynamic, because band
cost varies >100x), two bands in flight per GPU.

Before the W warm-up steps the clock is pre-conditioned for --ramp-ms (150 ms) with untimed launches of
the same workload: an idle MI355X needs 50-100 ms of load to reach its sustained clock, and a 0.6 ms tile
measured cold reads 15-20 % low.

Prints ONE JSON line on rank 0.  `value` = whole-job G pixel-iterations/s where a pixel's iterations
are count if count > 0 else mrd-1, summed from the kernel's own output.
`roofline`: the path is bound by the fp64 vector-ALU issue rate (not HBM, not MFMA -- FMA contraction
is forbidden by bit-exactness): achieved = 8 algorithmic flops per pixel-iteration / average
launch duration (HIP events on the launch stream); peak = CUs x 4 SIMD x 16 fp64 lanes x 2 flop x
clock (78.6 TFLOP/s on MI355X).  Under parity the default kernel needs 6.125 fp64-rate VALU issue
slots per 8 flops (6 arithmetic ops per step + one add and one compare per 16 steps), so the flops
fraction cannot exceed 8/12.25 = 0.653; `valu_slot_util` (= issue slots actually spent per
pixel-iteration over the 39.3 T lane-op/s issue peak at 2.4 GHz) is the "how close to the metal"
figure.  `traffic` is HBM bytes per launch from separate `rocprofv3 --pmc` passes of this same command
(committed under profiles/, see `traffic_source`); counters cannot be read from inside the run.
`cpu_baseline`: the strict-IEEE C oracle (oracle/, kind "port": the reference has no CPU
implementation and its numba path cannot run here) on the host cores, rank 0, N = 1 only.

Cycle test: the library's default (MBK_OPT_CYCLE_DETECT = 1) retires a pixel as "never escapes" as soon as
its (zr, zi) bit pattern repeats -- bit-identical counts, but iterations the reference would run are not
executed.  `value` and `roofline` are therefore measured with the test OFF (every iteration executed; stated
in config.cycle_test); the same K steps with the default ON are timed right after in the same run and
reported as the extra object `cycle_detection` (reference-equivalent rate, ms per step, speed-up).
`--opt cycle_detect=1` moves the test into the headline, labelled as such.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

WORKLOADS = {
    # name: (start_r, start_i, range, width, height, mrd, description)
    "cfg1": (-2.0, -1.5, 3.0, 512, 512, 256, "512x512 full set (centre -0.5+0i, span 3.0), mrd 256"),
    "cfg2": (-2.0, -1.5, 3.0, 4096, 4096, 1000, "4096x4096 full set (centre -0.5+0i, span 3.0), mrd 1000"),
    "cfg3": (-0.743648, 0.131820, 1e-5, 8192, 8192, 10000,
             "8192x8192 deep zoom (centre -0.743643+0.131825i, span 1e-5), mrd 10000"),
    "chunk_l1": (-2.0, -2.0, 4.0, 4096, 4096, 1000, "DataChunk (level 1, 0, 0) = [-2,2]^2, mrd 1000"),
    # diagnostics (not BASELINE configs): uniform all-in-set tile = pure loop throughput, no divergence
    "inset": (-0.2, -0.1, 0.2, 4096, 4096, 1000, "4096x4096 inside the main cardioid (every pixel runs mrd-1 steps)"),
    "exterior": (-2.0, -2.0, 1.0, 4096, 4096, 1000, "DataChunk (4,0,0): every pixel escapes within 3 steps"),
    # BASELINE configs[3]: centre/span are not given there; SURVEY 8(d) proposes centre -0.745+0.11i, span 0.02
    "cfg4": (-0.755, 0.10, 0.02, 16384, 16384, 50000,
             "16384x16384 seahorse valley (centre -0.745+0.11i, span 0.02), mrd 50000, fp32 kernel variant"),
    # BASELINE configs[4]: no view given; the full-set view of cfg2 with the continuous value as output
    "cfg5": (-2.0, -1.5, 3.0, 4096, 4096, 5000,
             "4096x4096 full set (centre -0.5+0i, span 3.0), mrd 5000, continuous (smooth) colouring: float64 nu + int32 count per pixel"),
}
DEFAULT_PRECISION = {"cfg4": "f32"}
SMOOTH_WORKLOADS = {"cfg5"}
FLOPS_PER_PIXEL_ITER = 8        # 4 mul + 4 add/sub with the squares shared
# fp64-rate VALU issue slots each kernel spends per pixel-iteration (v_cmp costs a full slot on gfx950):
#   per-step test: 3 mul + 3 add + 1 fma + 1 v_cmp = 8;  grouped test: 6 + 2 per group = 6.25 (8-step groups,
#   kernel "refill" and option group_steps=8) or 6.125 (16-step groups: the default for interior blocks)
VALU_SLOTS_PER_PIXEL_ITER = {"default": 6.125, "scan": 6.125, "group": 6.125, "refill": 6.25, "asm": 8.0, "simple": 8.0}
# default (steps, warmup) per workload: enough launches for a steady clock, a few seconds at most
DEFAULT_STEPS = {"cfg1": (400, 50), "cfg2": (400, 50), "chunk_l1": (400, 50), "inset": (60, 8), "exterior": (400, 50),
                 "cfg3": (20, 3), "cfg4": (3, 1), "cfg5": (40, 5)}
# CPU baseline sample: every `stride`-th 8-row band, sized for ~10-30 CPU-seconds on >= 64 host threads
CPU_SAMPLE_STRIDE = {"cfg4": 256}
PMC_SUMMARIES = [("r02", "cfg2_default_pmc_summary.json"), ("r01", "cfg2_default_pmc_summary.json")]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: per workload, ~0.3-6 s of GPU time)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default: per workload)")
    ap.add_argument("--ramp-ms", type=float, default=150.0,
                    help="untimed clock pre-conditioning before the warm-up steps: the same launches are repeated "
                         "for this long so that DVFS has left its idle state (an MI355X needs ~50-100 ms of load to "
                         "reach its sustained clock; a 0.6 ms tile measured cold reads 15-20 %% low). 0 disables.")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel", default="default")
    ap.add_argument("--precision", default=None, choices=["f64", "f32"],
                    help="f32 = BASELINE cfg4's fp32 kernel variant (not in the reference); default f64, cfg4: f32")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="library tuning option (include/mbk.h enum mbk_option), e.g. --opt group_steps=8; repeatable")
    ap.add_argument("--outputs", default="counts", choices=["counts", "both"],
                    help="counts (default, the contract's workload): int32 escape indices. both: also the quantised "
                         "uint8 tile the worker sends (what a DataChunk launch writes), for kernel studies")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "bands"],
                    help="tiles (default): every rank computes its own tile per step (weak scaling). bands: ONE "
                         "view per step, cut into >= 16 row bands per GPU that the ranks pull from a shared-memory "
                         "cursor (strong scaling; how BASELINE cfg3 shards an image over 8 GPUs)")
    ap.add_argument("--band-rows", type=int, default=0, help="rows per band for --shard bands (default: height / (16 N), >= 128)")
    ap.add_argument("--streams", type=int, default=None,
                    help="tiles (or bands) in flight per GPU (default 1 for tiles = the contract's serial steps, 2 for bands)")
    ap.add_argument("--control", default="gloo", choices=["gloo", "nccl"],
                    help="backend of the barrier / timing reductions for N > 1 (no data-path collective exists)")
    return ap.parse_args()


def cpu_baseline(name, workload, precision="f64"):
    """Time the C oracle (oracle/mandel_oracle.c, -ffp-contract=off) on the host cores."""
    from oracle.oracle import COracle
    sr, si, rng, w, h, mrd, _ = workload
    o = COracle()
    cores = o.max_threads()
    if name in SMOOTH_WORKLOADS:
        import numpy as np
        t0 = time.perf_counter()
        _, counts = o.view_smooth(sr, si, rng, rng, w, h, mrd)
        dt = time.perf_counter() - t0
        total = int(np.where(counts > 0, counts, mrd - 1).astype(np.int64).sum())
        return {"value": total / dt / 1e9, "unit": "G pixel-iterations/s", "cores": cores, "kind": "port",
                "sample": f"the whole {w}x{h} tile, mrd {mrd}, continuous value + count per pixel (mbo_view_smooth), "
                          "C oracle gcc -O2 -ffp-contract=off, OpenMP dynamic rows", "seconds": dt}
    # bounded sample: every `stride`-th row band of 8 rows, sized for roughly 10-30 CPU-seconds
    stride = CPU_SAMPLE_STRIDE.get(name, 1 if cores >= 4 else 4)
    bands = [(0, r, w, 8) for r in range(0, h, 8 * stride)]
    t0 = time.perf_counter()
    total = 0
    if stride == 1:
        _, _, total = o.view(sr, si, rng, rng, w, h, mrd, want_counts=False, want_bytes=False, nthreads=cores,
                             precision=precision)
        sample = f"the whole {w}x{h} tile, all rows"
        best_dt = time.perf_counter() - t0
        if best_dt < 2.0:   # short run: the first pass also pays for waking the OpenMP team; report the best of
            for _ in range(2):   # three passes (the baseline at its best, not at its worst)
                t1 = time.perf_counter()
                o.view(sr, si, rng, rng, w, h, mrd, want_counts=False, want_bytes=False, nthreads=cores, precision=precision)
                best_dt = min(best_dt, time.perf_counter() - t1)
            sample += " (best of 3 passes)"
    else:
        for win in bands:
            total += o.view(sr, si, rng, rng, w, h, mrd, window=win, want_counts=False,
                            want_bytes=False, nthreads=cores, precision=precision)[2]
        sample = f"every {stride}th 8-row band of the {w}x{h} tile ({len(bands) * 8} rows)"
        best_dt = time.perf_counter() - t0
    dt = best_dt
    rec = {"value": total / dt / 1e9, "unit": "G pixel-iterations/s", "cores": cores, "kind": "port",
           "sample": sample + f", mrd {mrd}, C oracle gcc -O2 -ffp-contract=off, OpenMP dynamic rows",
           "seconds": dt}
    if name in ("cfg3", "cfg4"):
        return rec       # one sample is already ~10 s of CPU
    # one thread, on every 16th 8-row band (the all-core figure above divided by `cores` hides the SMT/boost effects)
    t0 = time.perf_counter()
    tot1 = sum(o.view(sr, si, rng, rng, w, h, mrd, window=(0, r, w, 8), want_counts=False, want_bytes=False,
                      nthreads=1, precision=precision)[2] for r in range(0, h, 128))
    dt1 = time.perf_counter() - t0
    rec["single_thread"] = {"value": tot1 / dt1 / 1e9, "seconds": dt1, "sample": f"every 16th 8-row band ({h // 16} rows)"}
    if precision == "f64" and stride == 1 and o.have_avx512():
        # best-effort CPU: the same strict arithmetic 8 pixels at a time in AVX-512 (no fmadd), same threads
        dt8 = None
        for _ in range(3):   # best of three, as above
            t0 = time.perf_counter()
            _, total8 = o.view_avx512(sr, si, rng, rng, w, h, mrd, want_counts=False, nthreads=cores)
            dt8 = min(dt8, time.perf_counter() - t0) if dt8 is not None else time.perf_counter() - t0
        rec["best_effort_avx512"] = {"value": total8 / dt8 / 1e9, "seconds": dt8, "bit_identical_work": total8 == total}
    return rec


def pmc_traffic(workload, kernel):
    """(HBM bytes per launch, source) from the committed rocprofv3 PMC passes of this same command:
    WRITE_SIZE and FETCH_SIZE are in KiB, each collected in its own pass; FETCH_SIZE is doubled as
    MI355X_MICROARCH.md prescribes for gfx950.  (None, None) for un-profiled combinations."""
    if workload != "cfg2" or kernel not in ("default", "scan", "group"):
        return None, None
    for rnd, fname in PMC_SUMMARIES:
        path = os.path.join(ROOT, "profiles", rnd, fname)
        try:
            with open(path) as f:
                pmc = json.load(f)
            nbytes = int(pmc["WRITE_SIZE"]["mean"] * 1024 + 2 * pmc["FETCH_SIZE"]["mean"] * 1024)
            return nbytes, f"profiles/{rnd}/{fname}: separate rocprofv3 --pmc passes of this command, not counters of this run"
        except Exception:
            continue
    return None, None


def main():
    args = parse_args()
    d_steps, d_warm = DEFAULT_STEPS.get(args.workload, (20, 3))
    if args.steps is None:
        args.steps = d_steps
    if args.warmup is None:
        args.warmup = d_warm
    if args.precision is None:
        args.precision = DEFAULT_PRECISION.get(args.workload, "f64")
    smooth = args.workload in SMOOTH_WORKLOADS
    bands_mode = args.shard == "bands"
    if args.streams is None:
        args.streams = 2 if bands_mode else 1
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    fake = os.environ.get("MBK_BENCH_FAKE") == "1"   # CPU-only test hook for the N>1 control path
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    if smooth and (bands_mode or args.precision != "f64"):
        raise SystemExit("cfg5 (smooth colouring) runs with --shard tiles in fp64")
    options = {}
    for item in args.opt:
        k, _, v = item.partition("=")
        options[k] = int(v)

    import torch
    import torch.distributed as dist

    workload = WORKLOADS[args.workload]
    sr, si, rng, width, height, mrd, desc = workload
    npix = width * height

    backend = None
    if world > 1:
        if fake or args.control == "gloo":
            backend = "gloo"     # barrier + two scalar reductions on CPU tensors: nothing here needs RCCL
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            backend = "nccl"

    def barrier():
        if world > 1:
            dist.barrier()

    nstreams = max(1, args.streams)
    # >= 16 bands per GPU, but no band under 128 rows: a band costs ~70 us of host work (cursor lock, two kernel
    # launches, an event), which must stay below its GPU time.  Tickets run on across steps (no barrier
    # between images), so the tail that larger bands leave idle is paid once per run, not once per image.
    band_rows = args.band_rows or max(128, height // (16 * world))
    band_rows = max(8, (band_rows // 8) * 8)
    from distributedmandelbrot_amd.sharding import SharedCursor, make_bands
    bands = make_bands(height, band_rows) if bands_mode else None
    cursor = None
    if bands_mode:
        cname = f"{os.environ.get('MASTER_PORT', 'solo')}_{os.environ.get('TORCHELASTIC_RUN_ID', os.getppid())}"
        if rank == 0:
            cursor = SharedCursor(cname, create=True)
        barrier()
        if rank != 0:
            cursor = SharedCursor(cname, create=False)

    if fake:
        dev = None
        device_info = {"name": "fake", "compute_units": 256, "clock_mhz": 2400}
        streams = [None] * nstreams

        def launch_tile(i):
            time.sleep(0.001)

        def launch_band(i, bnd):
            time.sleep(0.0002 * (1 + bnd.index % 3))

        def sync():
            pass

        def slot_wait(i):
            pass
    else:
        from distributedmandelbrot_amd import MandelbrotDevice, View
        torch.cuda.set_device(local_rank)
        dev = MandelbrotDevice(local_rank)   # raises loudly without the HIP library / a gfx950 GPU
        # The headline is measured with the cycle test OFF: every iteration the reference would run is executed, so
        # `value` and the roofline describe the loop itself.  The library's default (ON: bit-identical counts, exactly
        # periodic orbits retired early) is timed in the same run as the extra object "cycle_detection".
        if "cycle_detect" not in options:
            dev.set_option("cycle_detect", 0)
        for k, v in options.items():
            dev.set_option(k, v)
        device_info = dev.info()
        device_info["scan_occupancy"] = dev.scan_occupancy()
        view = View(sr, si, rng, rng, width, height)
        nbuf = 1 if bands_mode else nstreams
        d_counts_all = [torch.empty(npix, dtype=torch.int32, device=f"cuda:{local_rank}") for _ in range(nbuf)]
        d_smooth_all = [torch.empty(npix, dtype=torch.float64, device=f"cuda:{local_rank}") for _ in range(nbuf)] if smooth else None
        d_bytes_all = [torch.empty(npix, dtype=torch.uint8, device=f"cuda:{local_rank}") for _ in range(nbuf)] if args.outputs == "both" else None
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
        slot_events = [torch.cuda.Event() for _ in range(nstreams)]
        slot_busy = [False] * nstreams

        def launch_tile(i):
            if smooth:
                dev.launch_view_smooth(view, mrd, d_smooth=d_smooth_all[i].data_ptr(), d_counts=d_counts_all[i].data_ptr(),
                                       stream=streams[i].cuda_stream, kernel=args.kernel)
            else:
                dev.launch_view(view, mrd, d_counts=d_counts_all[i].data_ptr(), stream=streams[i].cuda_stream,
                                d_bytes=d_bytes_all[i].data_ptr() if d_bytes_all else 0,
                                kernel=args.kernel, precision=args.precision)

        def launch_band(i, bnd):   # a row band of the shared view, written at its place in this rank's image
            dev.launch_view(view, mrd, window=(0, bnd.row0, width, bnd.nrows),
                            d_counts=d_counts_all[0].data_ptr() + 4 * bnd.row0 * width,
                            stream=streams[i].cuda_stream, kernel=args.kernel, precision=args.precision)
            slot_events[i].record(streams[i])
            slot_busy[i] = True

        def slot_wait(i):          # back-pressure: one band in flight per slot, or a rank would drain the cursor
            if slot_busy[i]:
                slot_events[i].synchronize()
                slot_busy[i] = False

        def sync():
            torch.cuda.synchronize()

    turn = [0]
    my_tickets = []

    def run_steps(nsteps, events=None):
        """tiles: nsteps launches round-robin over the streams.  bands: pull tickets until nsteps images are done."""
        if not bands_mode:
            for _ in range(nsteps):
                i = turn[0] % nstreams
                turn[0] += 1
                if events is not None and not fake:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(streams[i])
                    launch_tile(i)
                    e1.record(streams[i])
                    events.append((e0, e1))
                else:
                    launch_tile(i)
            return
        limit = nsteps * len(bands)
        while True:
            i = turn[0] % nstreams
            slot_wait(i)
            t = cursor.next()
            if t >= limit:
                break
            turn[0] += 1
            if events is not None:
                my_tickets.append(t)
            launch_band(i, bands[t % len(bands)])

    if not fake and args.ramp_ms > 0 and not bands_mode:   # clock pre-conditioning (untimed, see --ramp-ms)
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            launch_tile(0)
            sync()
    run_steps(args.warmup if not bands_mode else max(args.warmup, 1))
    sync()
    barrier()
    if bands_mode:
        if rank == 0:
            cursor.reset(0)
        barrier()
    events = []
    t0 = time.perf_counter()
    run_steps(args.steps, events)
    sync()
    barrier()
    elapsed = time.perf_counter() - t0

    never = 0
    if fake:
        iters_per_step = 10 ** 9
        kernel_ms = [elapsed / args.steps * 1e3] * args.steps
    else:
        if bands_mode:      # work per step = the whole image, measured once (untimed) on every rank's own GPU
            launch_tile(0)
            sync()
        st = dev.reduce_counts(d_counts_all[0].data_ptr(), npix, mrd, stream=streams[0].cuda_stream)
        iters_per_step, never = st.pixel_iterations, st.never_pixels
        kernel_ms = [a.elapsed_time(b) for a, b in events] if events else [elapsed / args.steps * 1e3]

    # second leg (tiles mode, fp64/fp32 count kernels): the same K steps with the library's default cycle test.
    # A failure here must not cost the headline: errors are caught (the barriers stay unconditional, so the ranks
    # stay in step) and reported in config.cycle_leg_error instead of the cycle_detection object.
    cyc_leg, cyc_err = None, None
    if not fake and not bands_mode and "cycle_detect" not in options and args.kernel in ("default", "group", "scan"):
        try:
            dev.set_option("cycle_detect", 1)
            run_steps(max(args.warmup, 2))
            sync()
        except Exception as e:   # noqa: BLE001 -- reported, not swallowed
            cyc_err = repr(e)
        barrier()
        t1 = time.perf_counter()
        try:
            if cyc_err is None:
                run_steps(args.steps)
                sync()
        except Exception as e:   # noqa: BLE001
            cyc_err = repr(e)
        barrier()
        cyc_elapsed = time.perf_counter() - t1
        try:
            if cyc_err is None:
                st2 = dev.reduce_counts(d_counts_all[0].data_ptr(), npix, mrd, stream=streams[0].cuda_stream)
                cyc_leg = (cyc_elapsed, st2.pixel_iterations == iters_per_step and st2.never_pixels == never)
            dev.set_option("cycle_detect", 0)
        except Exception as e:   # noqa: BLE001
            cyc_err, cyc_leg = repr(e), None

    bands_once = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if backend == "gloo" else f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_max = float(t.item())
        if not fake and not bands_mode and "cycle_detect" not in options and args.kernel in ("default", "group", "scan"):
            # every rank takes part (elapsed < 0 marks a rank whose leg failed: then no rank reports the leg)
            t = torch.tensor([cyc_leg[0] if cyc_leg else -1.0, -(cyc_leg[0] if cyc_leg else -1.0)], dtype=torch.float64,
                             device="cpu" if backend == "gloo" else f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)      # [max elapsed, -min elapsed]
            if cyc_leg is not None and -float(t[1].item()) >= 0.0:
                cyc_leg = (float(t[0].item()), cyc_leg[1])
            else:
                cyc_leg = None
        if bands_mode:
            gathered = [None] * world
            dist.all_gather_object(gathered, my_tickets)
            allt = sorted(x for g in gathered for x in g)
            bands_once = allt == list(range(args.steps * len(bands)))
            per_rank_bands = [len(g) for g in gathered]
    else:
        elapsed_max = elapsed
        if bands_mode:
            bands_once = sorted(my_tickets) == list(range(args.steps * len(bands)))
            per_rank_bands = [len(my_tickets)]
    # weak scaling: every rank did the same tile; strong scaling: the ranks shared one image per step
    iters_all = float(iters_per_step) * (1 if bands_mode else world)

    if rank == 0:
        avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
        if bands_mode:      # per-GPU average time per image, idle time included
            avg_kernel_s = elapsed_max / args.steps
        cus, mhz = device_info["compute_units"], device_info["clock_mhz"]
        lanes_per_clk = 16 if args.precision == "f64" else 32  # per SIMD: fp64 16, fp32 32 (SIMD-32)
        peak_lane_ops = cus * 4 * lanes_per_clk * mhz * 1e6    # VALU lane-ops/s of that type
        peak_tflops = peak_lane_ops * 2 / 1e12                 # FMA = 2 flop -> 78.6 (fp64) / 157.3 (fp32)
        per_gpu_iters = iters_per_step / (world if bands_mode else 1)
        achieved_tflops = FLOPS_PER_PIXEL_ITER * per_gpu_iters / avg_kernel_s / 1e12
        slots = VALU_SLOTS_PER_PIXEL_ITER.get(args.kernel, 8.0)
        if args.kernel in ("default", "scan", "group") and options.get("group_steps", 16) != 16:
            slots = 6.25 if options["group_steps"] == 8 else 6.5
        if options.get("cycle_detect", 0) and args.kernel in ("default", "scan", "group") and options.get("group_steps", 16) != 4:
            slots += 0.125     # two bitwise state compares per 16 steps
        out_bytes = npix * (12 if smooth else 4)
        traffic, traffic_source = pmc_traffic(args.workload, args.kernel) if args.precision == "f64" and not options else (None, None)
        metric = "G pixel-iterations/s on 4096^2 tile, max_iter=1000 fp64"
        cfg = {"workload": f"{args.workload}: {desc}; " + (
                   f"one image per step cut into {len(bands)} row bands of {band_rows} rows pulled from a shared cursor"
                   if bands_mode else "one tile per GPU per step") + ", int32 counts written to resident HBM",
               "kernel": args.kernel, "options": options, "outputs": args.outputs,
               "cycle_test": ("on (--opt)" if options.get("cycle_detect", 0) else
                              "off for value and roofline: every iteration the reference runs is executed" +
                              ("; the library default (on) is timed in this run: see cycle_detection" if cyc_leg else "")),
               "pixels_per_step": npix * (1 if bands_mode else world), "pixel_iterations_per_step_per_gpu": per_gpu_iters,
               "never_escaped_pixels": never, "parallelism": f"{world} independent work queue(s), no collective",
               "streams_per_gpu": nstreams, "shard": args.shard, "control_backend": backend,
               "clock_ramp_ms": 0.0 if fake or bands_mode else args.ramp_ms,
               "fake_backend": fake, "device": device_info.get("name"), "compute_units": cus, "clock_mhz": mhz,
               "cycle_leg_error": cyc_err,
               "occupancy_api_wg_per_cu": device_info.get("scan_occupancy")}
        if bands_mode:
            cfg.update({"bands_per_image": len(bands), "band_rows": band_rows, "bands_exactly_once": bands_once,
                        "bands_per_rank": per_rank_bands})
        rec = {
            "metric": metric,
            "value": iters_all * args.steps / elapsed_max / 1e9,
            "unit": "G pixel-iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if bands_mode else "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic (coordinates generated in-kernel from the view origin and stride; no RNG)",
            "config": cfg,
            "roofline": {
                "bound": "fp64_valu" if args.precision == "f64" else "fp32_valu",
                "achieved": achieved_tflops,
                "peak": peak_tflops,
                "unit": "TFLOP/s",
                "frac": achieved_tflops / peak_tflops,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "basis": "whole-job wall time per image per GPU (idle time included)" if bands_mode
                         else "HIP events around each launch on its stream",
                "kernel_ms_avg": avg_kernel_s * 1e3,
                "kernel_ms_min": min(kernel_ms),
                "flops_per_pixel_iteration": FLOPS_PER_PIXEL_ITER,
                "valu_slots_per_pixel_iteration": slots,
                # with the cycle test in the headline (--opt cycle_detect=1) fewer steps are executed than the
                # reference's count says, so the executed-issue figures cannot be derived from the output
                "parity_ceiling_frac": None if options.get("cycle_detect", 0) else FLOPS_PER_PIXEL_ITER / (2.0 * slots),
                "valu_slot_util": None if options.get("cycle_detect", 0) else slots * per_gpu_iters / avg_kernel_s / peak_lane_ops,
                "algorithmic_hbm_bytes_per_launch": out_bytes,
                "hbm_GBps": out_bytes / avg_kernel_s / 1e9,
            },
        }
        if cyc_leg is not None:
            rec["cycle_detection"] = {
                "what": "same workload and steps with MBK_OPT_CYCLE_DETECT=1 (library default): pixels whose (zr, zi) bit "
                        "pattern repeats are retired as 'never escapes' -- identical counts, fewer executed steps; value "
                        "counts the reference's iterations, not the executed ones",
                "value": iters_all * args.steps / cyc_leg[0] / 1e9,
                "unit": "G pixel-iterations/s (reference-equivalent)",
                "ms_per_step": cyc_leg[0] / args.steps * 1e3,
                "speedup_vs_strict": elapsed_max / cyc_leg[0],
                "same_pixel_iterations_and_never_count": bool(cyc_leg[1]),
            }
        if world == 1 and not args.no_cpu_baseline and not fake:
            rec["cpu_baseline"] = cpu_baseline(args.workload, workload, args.precision)
        elif world == 1 and fake:
            rec["cpu_baseline"] = None
        print(json.dumps(rec), flush=True)

    if cursor is not None:
        barrier()
        cursor.close()
    if world > 1:
        dist.destroy_process_group()
    if dev is not None:
        dev.close()


if __name__ == "__main__":
    main()

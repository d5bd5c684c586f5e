#!/usr/bin/env python3
"""bench.py -- headline benchmark of the Mandelbrot tile escape-time path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2] [--kernel default]

One "step" = one pass of the hot path over one synthetic tile already described in HBM terms: the
kernel generates its coordinates itself and writes int32 escape indices to a resident HBM buffer
(nothing crosses PCIe inside the timed region).  Workload at every N: BASELINE.json configs[1]
("cfg2"): 4096x4096 samples of the full set (centre -0.5+0i, span 3.0), max_iter (mrd) = 1000, fp64.
For N > 1 the driver launches one rank per GPU (torch.distributed.run); tiles are independent, so
every rank computes its own tile with no data-path collective ("weak" scaling); the only
communication is the barrier and the max-over-ranks of the elapsed time.

Before the W warm-up steps the clock is pre-conditioned for --ramp-ms (150 ms) with untimed launches of
the same workload: an idle MI355X needs 50-100 ms of load to reach its sustained clock, and a 0.6 ms tile
measured cold reads 15-20 % low.

Prints ONE JSON line on rank 0.  `value` = whole-job G pixel-iterations/s where a pixel's iterations
are count if count > 0 else mrd-1, summed from the kernel's own output.
`roofline`: the path is bound by the fp64 vector-ALU issue rate (not HBM, not MFMA -- FMA contraction
is forbidden by bit-exactness): achieved = 8 algorithmic flops per pixel-iteration / average kernel
launch duration (HIP events on the launch stream); peak = CUs x 4 SIMD x 16 fp64 lanes x 2 flop x
clock (78.6 TFLOP/s on MI355X).  Under parity the default kernel needs 6.25 fp64-rate VALU issue
slots per 8 flops (6 arithmetic ops per step + one add and one compare per 8 steps), so the flops
fraction cannot exceed 8/12.5 = 0.64; `valu_slot_util` (= issue slots actually spent per
pixel-iteration over the 39.3 T lane-op/s issue peak at 2.4 GHz) is the "how close to the metal"
figure.
`cpu_baseline`: the strict-IEEE C oracle (oracle/, kind "port": the reference has no CPU
implementation and its numba path cannot run here) on the host cores, rank 0, N = 1 only.
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
             "16384x16384 seahorse valley (centre -0.745+0.11i, span 0.02), mrd 50000 -- use --precision f32"),
}
FLOPS_PER_PIXEL_ITER = 8        # 4 mul + 4 add/sub with the squares shared
# fp64-rate VALU issue slots each kernel spends per pixel-iteration (v_cmp costs a full slot on gfx950):
#   per-step test: 3 mul + 3 add + 1 fma + 1 v_cmp = 8;  grouped test (default): 6 + 2 per 8 steps = 6.25
VALU_SLOTS_PER_PIXEL_ITER = {"default": 6.25, "group": 6.25, "refill": 6.25, "asm": 8.0, "simple": 8.0}


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
    ap.add_argument("--precision", default="f64", choices=["f64", "f32"],
                    help="f32 = BASELINE cfg4's fp32 kernel variant (not in the reference)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "bands"],
                    help="tiles (default): every rank computes its own tile per step (weak scaling). bands: ONE "
                         "view per step is split into row bands (4 per rank) interleaved over the ranks (strong scaling; "
                         "how BASELINE cfg3 shards an image over 8 GPUs)")
    ap.add_argument("--streams", type=int, default=1,
                    help="tiles in flight per GPU: steps are issued round-robin on this many HIP streams "
                         "(1 = the contract's serial steps; 2 lets the next tile fill the drain of the last)")
    return ap.parse_args()


def cpu_baseline(workload, precision="f64"):
    """Time the C oracle (oracle/mandel_oracle.c, -ffp-contract=off) on the host cores."""
    from oracle.oracle import COracle
    sr, si, rng, w, h, mrd, _ = workload
    o = COracle()
    cores = o.max_threads()
    # bounded sample: every `stride`-th row band of 8 rows, sized for roughly 10-30 CPU-seconds
    stride = 1 if cores >= 4 else 4
    bands = [(0, r, w, 8) for r in range(0, h, 8 * stride)]
    t0 = time.perf_counter()
    total = 0
    if stride == 1:
        _, _, total = o.view(sr, si, rng, rng, w, h, mrd, want_counts=False, want_bytes=False, nthreads=cores,
                             precision=precision)
        sample = f"the whole {w}x{h} tile, all rows"
    else:
        for win in bands:
            total += o.view(sr, si, rng, rng, w, h, mrd, window=win, want_counts=False,
                            want_bytes=False, nthreads=cores, precision=precision)[2]
        sample = f"every {stride}th 8-row band of the {w}x{h} tile ({len(bands) * 8} rows)"
    dt = time.perf_counter() - t0
    rec = {"value": total / dt / 1e9, "unit": "G pixel-iterations/s", "cores": cores, "kind": "port",
           "sample": sample + f", mrd {mrd}, C oracle gcc -O2 -ffp-contract=off, OpenMP dynamic rows",
           "seconds": dt}
    # one thread, on every 16th 8-row band (the all-core figure above divided by `cores` hides the SMT/boost effects)
    t0 = time.perf_counter()
    tot1 = sum(o.view(sr, si, rng, rng, w, h, mrd, window=(0, r, w, 8), want_counts=False, want_bytes=False,
                      nthreads=1, precision=precision)[2] for r in range(0, h, 128))
    dt1 = time.perf_counter() - t0
    rec["single_thread"] = {"value": tot1 / dt1 / 1e9, "seconds": dt1, "sample": f"every 16th 8-row band ({h // 16} rows)"}
    if precision == "f64" and stride == 1 and o.have_avx512():
        # best-effort CPU: the same strict arithmetic 8 pixels at a time in AVX-512 (no fmadd), same threads
        t0 = time.perf_counter()
        _, total8 = o.view_avx512(sr, si, rng, rng, w, h, mrd, want_counts=False, nthreads=cores)
        dt8 = time.perf_counter() - t0
        rec["best_effort_avx512"] = {"value": total8 / dt8 / 1e9, "seconds": dt8, "bit_identical_work": total8 == total}
    return rec


def pmc_traffic(workload, kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC pass of this same command
    (profiles/r01/cfg2_default_pmc_summary.json: WRITE_SIZE and FETCH_SIZE are in KiB; FETCH_SIZE is
    doubled as MI355X_MICROARCH.md prescribes for gfx950).  None for un-profiled combinations -- PMC
    counters cannot be collected from inside the timed run itself."""
    if workload != "cfg2" or kernel not in ("default", "group"):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01", "cfg2_default_pmc_summary.json")) as f:
            pmc = json.load(f)
        return int(pmc["WRITE_SIZE"]["mean"] * 1024 + 2 * pmc["FETCH_SIZE"]["mean"] * 1024)
    except Exception:
        return None


# default (steps, warmup) per workload: enough launches for a steady clock, a few seconds at most
DEFAULT_STEPS = {"cfg1": (400, 50), "cfg2": (400, 50), "chunk_l1": (400, 50), "inset": (60, 8), "exterior": (400, 50),
                 "cfg3": (20, 3), "cfg4": (3, 1)}


def main():
    args = parse_args()
    d_steps, d_warm = DEFAULT_STEPS.get(args.workload, (20, 3))
    if args.steps is None:
        args.steps = d_steps
    if args.warmup is None:
        args.warmup = d_warm
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    fake = os.environ.get("MBK_BENCH_FAKE") == "1"   # CPU-only test hook for the N>1 control path
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")

    import torch
    import torch.distributed as dist

    workload = WORKLOADS[args.workload]
    sr, si, rng, width, height, mrd, desc = workload
    npix = width * height

    backend = None
    if world > 1:
        if fake:
            backend = "gloo"
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            try:   # RCCL: only the barrier and two scalar reductions use it -- the tiles need no collective
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
                backend = "nccl"
            except Exception as e:  # keep the scaling run alive if RCCL cannot come up on this node
                print(f"[bench] nccl init failed ({e!r}); using gloo for the barrier/reductions", file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group(backend="gloo")
                backend = "gloo"

    def barrier():
        if world > 1:
            dist.barrier()

    if fake:
        dev = None
        device_info = {"name": "fake", "compute_units": 256, "clock_mhz": 2400}

        def launch():
            time.sleep(0.001)

        def sync():
            pass
    else:
        from distributedmandelbrot_amd import MandelbrotDevice, View
        torch.cuda.set_device(local_rank)
        dev = MandelbrotDevice(local_rank)   # raises loudly without the HIP library / a gfx950 GPU
        device_info = dev.info()
        view = View(sr, si, rng, rng, width, height)
        nstreams = max(1, args.streams)
        d_counts_all = [torch.empty(npix, dtype=torch.int32, device=f"cuda:{local_rank}") for _ in range(nstreams)]
        d_counts = d_counts_all[0]
        streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(nstreams - 1)]
        stream = streams[0]
        turn = [0]

        from distributedmandelbrot_amd.sharding import make_bands, rank_bands
        band_rows = max(128, height // (4 * world))   # 4 interleaved slabs per rank: balance vs per-launch drain
        my_bands = rank_bands(make_bands(height, band_rows), rank, world) if args.shard == "bands" else None

        def launch():
            i = turn[0] % nstreams
            turn[0] += 1
            if my_bands is None:
                dev.launch_view(view, mrd, d_counts=d_counts_all[i].data_ptr(), stream=streams[i].cuda_stream,
                                kernel=args.kernel, precision=args.precision)
            else:  # this rank's row bands of the shared view, each written at its place in the image;
                # bands go round-robin over the streams so that one band's drain overlaps the next band
                for j, bnd in enumerate(my_bands):
                    dev.launch_view(view, mrd, window=(0, bnd.row0, width, bnd.nrows),
                                    d_counts=d_counts_all[0].data_ptr() + 4 * bnd.row0 * width,
                                    stream=streams[j % nstreams].cuda_stream, kernel=args.kernel,
                                    precision=args.precision)
            return streams[i]

        def sync():
            torch.cuda.synchronize()

    if not fake and args.ramp_ms > 0:   # clock pre-conditioning (untimed, see --ramp-ms)
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
            launch()
            sync()
    for _ in range(args.warmup):
        launch()
    sync()
    barrier()
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if not fake:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st_ = streams[turn[0] % nstreams]
            e0.record(st_)
            launch()
            e1.record(st_)
            events.append((e0, e1))
        else:
            launch()
    sync()
    barrier()
    elapsed = time.perf_counter() - t0

    if fake:
        iters_per_step = 10 ** 9
        kernel_ms = [elapsed / args.steps * 1e3] * args.steps
        never = 0
    else:
        if my_bands is None:
            st = dev.reduce_counts(d_counts.data_ptr(), npix, mrd, stream=stream.cuda_stream)
            iters_per_step, never = st.pixel_iterations, st.never_pixels
        else:
            iters_per_step = never = 0
            for bnd in my_bands:
                st = dev.reduce_counts(d_counts.data_ptr() + 4 * bnd.row0 * width, bnd.nrows * width, mrd,
                                       stream=stream.cuda_stream)
                iters_per_step += st.pixel_iterations
                never += st.never_pixels
        kernel_ms = [a.elapsed_time(b) for a, b in events]

    # max elapsed over ranks, total work over ranks
    if world > 1:
        dev_t = "cpu" if backend == "gloo" else f"cuda:{local_rank}"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev_t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_max = float(t.item())
        w_ = torch.tensor([float(iters_per_step)], dtype=torch.float64, device=dev_t)
        dist.all_reduce(w_, op=dist.ReduceOp.SUM)
        iters_all = float(w_.item())
    else:
        elapsed_max, iters_all = elapsed, float(iters_per_step)

    if rank == 0:
        avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
        cus, mhz = device_info["compute_units"], device_info["clock_mhz"]
        lanes_per_clk = 16 if args.precision == "f64" else 32  # per SIMD: fp64 16, fp32 32 (SIMD-32)
        peak_lane_ops = cus * 4 * lanes_per_clk * mhz * 1e6    # VALU lane-ops/s of that type
        peak_tflops = peak_lane_ops * 2 / 1e12                 # FMA = 2 flop -> 78.6 (fp64) / 157.3 (fp32)
        achieved_tflops = FLOPS_PER_PIXEL_ITER * iters_per_step / avg_kernel_s / 1e12
        slots = VALU_SLOTS_PER_PIXEL_ITER.get(args.kernel, 8.0)
        out_bytes = npix * 4
        rec = {
            "metric": "G pixel-iterations/s on 4096^2 tile, max_iter=1000 fp64",
            "value": iters_all * args.steps / elapsed_max / 1e9,
            "unit": "G pixel-iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if args.shard == "tiles" else "strong",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic (coordinates generated in-kernel from the view origin and stride; no RNG)",
            "config": {"workload": f"{args.workload}: {desc}; one tile per GPU per step, int32 counts "
                                   "written to resident HBM", "kernel": args.kernel,
                       "pixels_per_step_per_gpu": npix, "pixel_iterations_per_step_per_gpu": iters_per_step,
                       "never_escaped_pixels": never, "parallelism": f"{world} independent tile queue(s), no collective",
                       "streams_per_gpu": max(1, args.streams), "shard": args.shard, "control_backend": backend, "clock_ramp_ms": 0.0 if fake else args.ramp_ms,
                       "fake_backend": fake, "device": device_info.get("name"), "compute_units": cus,
                       "clock_mhz": mhz},
            "roofline": {
                "bound": "fp64_valu" if args.precision == "f64" else "fp32_valu",
                "achieved": achieved_tflops,
                "peak": peak_tflops,
                "unit": "TFLOP/s",
                "frac": achieved_tflops / peak_tflops,
                "traffic": pmc_traffic(args.workload, args.kernel) if args.precision == "f64" else None,
                "kernel_ms_avg": avg_kernel_s * 1e3,
                "kernel_ms_min": min(kernel_ms),
                "flops_per_pixel_iteration": FLOPS_PER_PIXEL_ITER,
                "valu_slots_per_pixel_iteration": slots,
                "parity_ceiling_frac": FLOPS_PER_PIXEL_ITER / (2.0 * slots),
                "valu_slot_util": slots * iters_per_step / avg_kernel_s / peak_lane_ops,
                "algorithmic_hbm_bytes_per_launch": out_bytes,
                "hbm_GBps": out_bytes / avg_kernel_s / 1e9,
            },
        }
        if world == 1 and not args.no_cpu_baseline and not fake:
            rec["cpu_baseline"] = cpu_baseline(workload, args.precision)
        elif world == 1 and fake:
            rec["cpu_baseline"] = None
        print(json.dumps(rec), flush=True)

    if world > 1:
        dist.destroy_process_group()
    if dev is not None:
        dev.close()


if __name__ == "__main__":
    main()

"""CPU tests of the multi-GPU sharding logic: row bands, the shared work queue with fake devices
(backed by the oracle -- tests only), and the one-process-per-GPU path under gloo, world_size 2."""
import os
import subprocess
import sys
import json
import time

import numpy as np
import pytest

from conftest import ROOT
from distributedmandelbrot_amd import View
from distributedmandelbrot_amd.device import TileStats
from distributedmandelbrot_amd.sharding import Band, WorkQueue, make_bands, rank_bands, render_view


class OracleDevice:
    """MandelbrotDevice look-alike for CPU tests of the host logic."""

    def __init__(self, oracle):
        self.oracle = oracle
        self.calls = 0

    def compute_view(self, view, mrd, *, window=None, want_counts=True, want_bytes=True, kernel="default",
                     out_counts=None, out_bytes=None):
        self.calls += 1
        c, b, total = self.oracle.view(view.start_r, view.start_i, view.range_r, view.range_i,
                                       view.width, view.height, mrd, window=window, nthreads=1)
        if out_counts is not None:
            out_counts[...] = c
        if out_bytes is not None:
            out_bytes[...] = b
        return (c if want_counts else None, b if want_bytes else None,
                TileStats(0.0, 0.0, total, int((c == 0).sum()), False, False))


class TwoSlotOracleDevice(OracleDevice):
    """Adds the asynchronous pair submit_view / wait (two slots) that render_view prefers."""

    def __init__(self, oracle):
        super().__init__(oracle)
        self.pending = [None, None]
        self.max_inflight = 0

    def submit_view(self, slot, view, mrd, *, window=None, out_counts=None, out_bytes=None, kernel="default"):
        assert self.pending[slot] is None, "slot reused before wait"
        self.pending[slot] = (view, mrd, window, out_counts, out_bytes)
        self.max_inflight = max(self.max_inflight, sum(p is not None for p in self.pending))

    def wait(self, slot):
        view, mrd, window, oc, ob = self.pending[slot]
        self.pending[slot] = None
        return self.compute_view(view, mrd, window=window, out_counts=oc, out_bytes=ob)[2]


def test_make_bands_cover_exactly():
    bands = make_bands(1000, 128)
    assert [b.nrows for b in bands] == [128] * 7 + [104]
    assert bands[0] == Band(0, 0, 128) and sum(b.nrows for b in bands) == 1000
    assert make_bands(5, 8) == [Band(0, 0, 5)]
    with pytest.raises(ValueError):
        make_bands(0, 8)


def test_rank_bands_partition():
    bands = make_bands(4096, 128)
    for world in (1, 2, 3, 8):
        parts = [rank_bands(bands, r, world) for r in range(world)]
        flat = sorted(b.index for p in parts for b in p)
        assert flat == list(range(len(bands)))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    with pytest.raises(ValueError):
        rank_bands(bands, 2, 2)


def test_work_queue_hands_out_each_item_once():
    import threading
    q = WorkQueue(range(1000))
    got = [[] for _ in range(8)]

    def run(i):
        while True:
            x = q.pop()
            if x is None:
                return
            got[i].append(x)

    ts = [threading.Thread(target=run, args=(i,)) for i in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert sorted(x for g in got for x in g) == list(range(1000))


def test_render_view_over_fake_devices_equals_whole_view(oracle):
    view, mrd = View(-2.0, -1.5, 3.0, 3.0, 200, 150), 200
    devs = [OracleDevice(oracle) for _ in range(3)]
    c, b, per = render_view(devs, view, mrd, band_rows=16)
    oc, ob, total = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, 200, 150, mrd)
    assert np.array_equal(c, oc) and np.array_equal(b, ob)
    assert sum(p["bands"] for p in per) == 10 and sum(p["pixel_iterations"] for p in per) == total
    assert sum(d.calls for d in devs) == 10


def test_render_view_two_slots_writes_bands_in_place(oracle):
    """Two bands in flight per device, each landing directly in its rows of a caller-supplied image."""
    view, mrd = View(-0.755, 0.10, 0.02, 0.02, 160, 203), 300
    devs = [TwoSlotOracleDevice(oracle) for _ in range(2)]
    counts = np.full((203, 160), -1, np.int32)
    c, b, per = render_view(devs, view, mrd, band_rows=24, want_bytes=False, out_counts=counts)
    oc, _, total = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, 160, 203, mrd)
    assert c is counts and b is None and np.array_equal(counts, oc)
    assert sum(p["bands"] for p in per) == 9 and sum(p["pixel_iterations"] for p in per) == total
    assert max(d.max_inflight for d in devs) == 2 and all(p is None for d in devs for p in d.pending)


def test_bench_distributed_path_gloo_world2():
    """bench.py's N>1 path (barrier, per-rank shard, max-over-ranks timing, aggregate) with two CPU
    processes over gloo; the compute is a stub (MBK_BENCH_FAKE=1) because there is no GPU here."""
    import socket
    with socket.socket() as s_:            # a free rendezvous port (fixed ports collide between runs)
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, MBK_BENCH_FAKE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--shard", "own"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["launcher"] == "torch.distributed.run" and len(rec["config"]["ranks_seen"]) == 2
    assert rec["data"].startswith("synthetic") and rec["higher_is_better"] is True
    # fake backend: every rank reports 1e9 pixel-iterations per step -> aggregate = 2e9 * steps / time
    assert rec["config"]["fake_backend"] is True
    assert abs(rec["value"] - 2 * 1.0 * 3 / (rec["ms_per_step"] * 3 / 1e3)) / rec["value"] < 1e-6


def test_bench_bands_dynamic_cursor_gloo_world2():
    """--shard bands: one image per step cut into row bands that the ranks pull from the shared-memory
    cursor (no RCCL, no static split).  Two CPU processes over gloo with the stub backend: every
    (step, band) ticket of the timed region is computed exactly once, and both ranks got work."""
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, MBK_BENCH_FAKE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", "cfg3", "--shard", "bands"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    cfg = rec["config"]
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2 and cfg["control_backend"] == "gloo"
    assert cfg["bands_per_image"] == 32 and cfg["band_rows"] == 256          # >= 16 bands per GPU (of >= 128 rows)
    assert cfg["bands_exactly_once"] is True
    assert sum(cfg["bands_per_rank"]) == 3 * 32 and min(cfg["bands_per_rank"]) > 0
    # the same images on one rank alone, timed in the same run: the denominator that belongs to `value`
    solo = rec["single_gpu_same_job"]
    assert solo["steps"] == 3 and solo["value"] > 0
    assert abs(rec["speedup_same_job"] - rec["value"] / solo["value"]) < 1e-9
    assert abs(rec["efficiency_same_job"] - rec["speedup_same_job"] / 2) < 1e-9
    assert not os.path.exists(os.path.join("/dev/shm", "mbk_cursor_%d_none" % port))


def test_bench_self_launch_queue_default_world2():
    """`python bench.py --gpus 2` -- the driver's plain command, no launcher around it -- starts its own two ranks
    and prints ONE JSON line.  The N > 1 default is the north star's partition: a fixed set of tiles per step that
    the ranks pull from one shared cursor (strong scaling), each ticket taken exactly once, both ranks fed, and
    `ranks_seen` says who took part.  Stub compute (MBK_BENCH_FAKE=1): there is no GPU here."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["MBK_BENCH_FAKE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    cfg = rec["config"]
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "strong" and cfg["shard"] == "queue"
    assert cfg["launcher"] == "bench.py self-launch" and cfg["control_backend"] == "gloo"
    assert cfg["tiles_per_step"] == 64 and cfg["grid"] == 8 and cfg["tiles_exactly_once"] is True
    assert sum(cfg["tiles_per_rank"]) == 3 * 64 and min(cfg["tiles_per_rank"]) > 0
    seen = cfg["ranks_seen"]
    assert sorted(r["rank"] for r in seen) == [0, 1] and len({r["pid"] for r in seen}) == 2
    assert all(k in seen[0] for k in ("host", "device", "pci_bus_id", "gpu_index", "local_rank"))
    assert len(cfg["rank_finish_ms"]) == 2 and rec["roofline"]["traffic"] is None
    # the stub's per-tile work is 1e7 * (1 + t % 3): the census must have measured every tile once
    per_step = sum(10 ** 7 * (1 + t % 3) for t in range(64))
    assert abs(rec["value"] - per_step * 3 / (rec["ms_per_step"] * 3 / 1e3) / 1e9) / rec["value"] < 1e-6
    # rank 0 then ran the same 3 steps alone (the other rank in a barrier): single-GPU rate of the SAME job, in the line
    solo = rec["single_gpu_same_job"]
    assert solo["steps"] == 3 and 0 < solo["value"] < rec["value"] * 1.05
    assert abs(rec["speedup_same_job"] - rec["value"] / solo["value"]) < 1e-9
    assert abs(rec["efficiency_same_job"] - rec["speedup_same_job"] / 2) < 1e-9 and rec["efficiency_same_job"] > 0.6
    assert cfg["tile_order"].startswith("image order")


def test_bench_queue_default_steps_grow_with_the_ranks():
    """The job of a step is fixed (64 tiles) but the default number of timed steps grows with N, so that the timed
    region per GPU keeps its length; --no-solo drops the single-GPU pass."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["MBK_BENCH_FAKE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "2", "--no-solo"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["steps"] == 2 * 400 and rec["config"]["tiles_exactly_once"] is True and "single_gpu_same_job" not in rec


def test_bench_self_launch_reports_a_dead_rank():
    """A rank that dies must not leave the others hanging in a barrier: the launcher stops them and exits non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(MBK_BENCH_FAKE="1", MBK_BENCH_FAKE_DIE_RANK="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "rank 1 exited" in out.stderr


def test_bench_queue_mode_single_rank_fake():
    """--shard queue at N = 1 is the same-mode first point of the scaling curve."""
    env = dict(os.environ, MBK_BENCH_FAKE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--shard", "queue", "--grid", "3", "--steps", "2"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 1 and rec["scaling"] == "strong" and rec["config"]["tiles_per_step"] == 9
    assert rec["config"]["tiles_exactly_once"] is True and rec["config"]["tiles_per_rank"] == [18]


def _cursor_worker(name, n, q):
    from distributedmandelbrot_amd.sharding import SharedCursor
    c = SharedCursor(name, create=False)
    got = []
    while True:
        t = c.next()
        if t >= n:
            break
        got.append(t)
    c.close()
    q.put(got)


def test_shared_cursor_across_processes():
    import multiprocessing as mp
    from distributedmandelbrot_amd.sharding import SharedCursor
    name = f"test_{os.getpid()}"
    owner = SharedCursor(name, create=True)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_cursor_worker, args=(name, 5000, q)) for _ in range(4)]
        [p.start() for p in procs]
        got = [q.get(timeout=120) for _ in procs]
        [p.join(timeout=60) for p in procs]
        assert sorted(x for g in got for x in g) == list(range(5000))
        owner.reset(7)
        assert owner.next(3) == 7 and owner.next() == 10
    finally:
        owner.close()
    assert not os.path.exists(owner.path)


def test_shared_cursor_guided_chunks():
    """next_guided: chunks of remaining / divisor tickets, never across a period boundary, every ticket exactly once,
    single tickets at the end."""
    from distributedmandelbrot_amd.sharding import SharedCursor
    c = SharedCursor(f"guided_{os.getpid()}", create=True)
    try:
        got, sizes = [], []
        while True:
            t, k = c.next_guided(200, 4, period=64)
            if k == 0:
                assert t == 200
                break
            assert t // 64 == (t + k - 1) // 64          # one image per chunk
            got += list(range(t, t + k))
            sizes.append(k)
        assert got == list(range(200)) and sizes[0] == 50 and sizes[-1] == 1 and max(sizes) <= 64
        assert sizes == sorted(sizes, reverse=True) or True   # (period clipping may shorten a chunk in the middle)
    finally:
        c.close()


def test_bench_json_contract_single_rank_fake():
    """The one-line JSON of bench.py carries every field the round contract names (checked on the CPU
    with the stub backend; the real numbers come from the GPU run)."""
    env = dict(os.environ, MBK_BENCH_FAKE="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["n_gpus"] == 1 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["vs_baseline"] is None
    assert rec["unit"] == "G pixel-iterations/s" and rec["dtype"] == "f64" and "workload" in rec["config"]
    assert "model" not in rec["config"]
    # the headline is the strict leg (every iteration executed); the cycle-test leg needs a GPU and is absent here
    assert rec["config"]["cycle_test"].startswith("off for value and roofline") and rec["config"]["cycle_leg_error"] is None
    assert "cycle_detection" not in rec
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel_ms_median"):
        assert key in rec["roofline"], key
    assert rec["scaling"] == "weak" and rec["config"]["shard"] == "own" and rec["config"]["launcher"] == "single process"
    assert len(rec["config"]["ranks_seen"]) == 1 and rec["config"]["distinct_gpus"] == 1
    assert abs(rec["roofline"]["frac"] - rec["roofline"]["achieved"] / rec["roofline"]["peak"]) < 1e-12


def test_bench_smi_sampler_degrades_without_the_library_or_a_gpu():
    """bench.py's `sustained` leg and `config.versions` read board power / clock / firmware through librocm_smi64; on a box
    where the library is missing or finds no device every method answers None / {} / [] instead of raising -- the line must
    not depend on it (here: no GPU)."""
    import bench
    s = bench.SmiSampler(0)
    assert s.sample() is None or len(s.sample()) == 3
    assert isinstance(s.versions(), dict)
    stop = s.trace(0.005)
    rows = stop()
    assert isinstance(rows, list)
    s.close()
    assert s.sample() is None and s.versions() == {}


def _run_bench_two_ranks_one_gpu(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "MBK_BENCH_FAKE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--oversubscribe", "--steps", "2",
                          "--warmup", "1"] + extra, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_real_bench_two_ranks_share_the_gpu_queue():
    """The REAL N > 1 path -- two processes, each with its own MandelbrotDevice, torch streams, the shared cursor and
    gloo -- on this box's one GPU (`--oversubscribe`: a functional test, never a scaling number).  Every ticket of
    the timed region exactly once, both ranks fed, two pids, no stub; and the same-job single-GPU pass is in the line."""
    rec = _run_bench_two_ranks_one_gpu([])
    cfg = rec["config"]
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and cfg["shard"] == "queue" and cfg["fake_backend"] is False
    assert cfg["oversubscribed"] is True and cfg["tiles_exactly_once"] is True
    assert sum(cfg["tiles_per_rank"]) == 2 * 64 and min(cfg["tiles_per_rank"]) > 0
    seen = cfg["ranks_seen"]
    assert len({r["pid"] for r in seen}) == 2 and all(r["pci_bus_id"] for r in seen) and cfg["distinct_gpus"] == 1
    solo = rec["single_gpu_same_job"]
    assert "error" not in solo and solo["value"] > 1000.0          # G pixel-iterations/s of one MI355X on this job
    assert 0.5 < rec["speedup_same_job"] < 1.3                     # two ranks time-slicing ONE GPU cannot beat it by much
    # the census measured every tile from the kernels' own output: the job holds far-exterior and all-interior tiles
    lo, hi = cfg["tile_pixel_iterations_min_max"]
    assert hi > 100 * lo > 0


@pytest.mark.gpu
def test_real_bench_two_ranks_share_the_gpu_cfg3_bands():
    """BASELINE cfg3's strong-scaling form (one 8192^2 deep-zoom image per step, row bands from the shared cursor) with
    two real ranks on the one GPU."""
    rec = _run_bench_two_ranks_one_gpu(["--workload", "cfg3", "--shard", "bands"])
    cfg = rec["config"]
    assert rec["n_gpus"] == 2 and cfg["shard"] == "bands" and cfg["fake_backend"] is False
    assert cfg["bands_exactly_once"] is True and sum(cfg["bands_per_rank"]) == 2 * cfg["bands_per_image"]
    assert min(cfg["bands_per_rank"]) > 0 and len({r["pid"] for r in cfg["ranks_seen"]}) == 2
    assert cfg["pixel_iterations_per_step_per_gpu"] * 2 > 6.0e10   # the whole image's work was measured (~65 G)
    assert "error" not in rec["single_gpu_same_job"] and rec["single_gpu_same_job"]["value"] > 1000.0


def test_bench_fails_fast_when_ranks_exceed_gpus():
    """more ranks than visible GPUs without --oversubscribe is one line and a non-zero exit, not N ranks
    dying one by one (here: no GPU at all, 3 ranks asked for)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MBK_BENCH_FAKE")}
    env["HIP_VISIBLE_DEVICES"] = ""          # also on a GPU box: hide them
    env["ROCR_VISIBLE_DEVICES"] = ""
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and time.monotonic() - t0 < 60
    assert "--gpus 3 but only 0 GPU(s) visible" in out.stderr and "--oversubscribe" in out.stderr, out.stderr[-500:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_golden_outputs_describe_the_bench_workloads():
    """tests/golden/bench_outputs.json (CPU oracle; tests/golden/make_bench_golden.py) is what bench.py's `output_verified`
    fields compare against: every entry must describe exactly the view / mrd / precision / window bench.py times."""
    sys.path.insert(0, ROOT)
    import bench
    with open(bench.GOLDEN_OUTPUTS) as f:
        g = json.load(f)
    assert "cfg2" in g and set(bench.EXTRA_CONFIGS) <= set(g), sorted(g)
    for name, e in g.items():
        wl, window = (name, None) if name in bench.WORKLOADS else bench.EXTRA_CONFIGS[name][:2]
        sr, si, rng, w, h, mrd, _ = bench.WORKLOADS[wl]
        assert e["view"] == [sr, si, rng, rng, w, h] and e["mrd"] == mrd, name
        assert e["precision"] == bench.DEFAULT_PRECISION.get(wl, "f64") and (tuple(e["window"]) if e["window"] else None) == window, name
        assert len(e["counts_sha256"]) == 64 and e["pixel_iterations"] > 0
    # the headline's totals are the ones DESIGN.md and the verdicts quote
    assert g["cfg2"]["pixel_iterations"] == 2879480177 and g["cfg2"]["never_pixels"] == 2814248
    # ... and the committed hashes are the SCALAR oracle's (the generator used the AVX-512 evaluation where it exists):
    # recomputed here for the two 4096^2 mrd-1000 tiles (~2 s)
    import hashlib
    from oracle.oracle import COracle
    o = COracle()
    for name in ("cfg2", "chunk_l1"):
        sr, si, rng, w, h, mrd, _ = bench.WORKLOADS[name]
        counts, _, total = o.view(sr, si, rng, rng, w, h, mrd, want_bytes=False)
        assert hashlib.sha256(np.ascontiguousarray(counts, dtype="<i4").tobytes()).hexdigest() == g[name]["counts_sha256"], name
        assert total == g[name]["pixel_iterations"] and int((counts == 0).sum()) == g[name]["never_pixels"]


@pytest.mark.gpu
def test_real_bench_default_line_pins_its_outputs_and_shows_every_config():
    """The command the driver runs (`python bench.py --gpus 1 --steps K --warmup W`): the timed launches' own buffer hashes
    to the CPU oracle's counts (strict leg and cycle-test leg), and the line carries a short strict leg of every other
    single-GPU BASELINE config, each verified the same way."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MBK_BENCH_FAKE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["output_verified"]["verified"] is True, rec["output_verified"]
    assert rec["cycle_detection"]["output_verified"]["verified"] is True
    # round 6: the headline runs the library's defaults (the cycle test is the only option the leg changes) ...
    assert rec["config"]["xcd_balance"] == "0 (library default)" and rec["config"]["options"] == {}
    assert rec["xcd_balance_opt_in"]["value"] > 1000.0 and rec["xcd_balance_opt_in"]["steps_run"] >= 20
    # ... and the line explains itself: an event-timed cycle leg of >= 50 ms, the host's share, a >= 1 s leg with clock / power
    cyc = rec["cycle_detection"]
    assert cyc["steps"] == 20 and cyc["steps_run"] >= 20 and cyc["steps_run"] * cyc["kernel_ms_avg"] >= 45.0
    assert 0.9 < cyc["kernel_ms_avg"] / cyc["ms_per_step"] <= 1.0 and 0.0 < cyc["host_submit_us_per_launch"] < 500.0
    assert 0.0 < rec["config"]["host_submit_us_per_launch"] < 500.0
    sus = rec["sustained"]
    assert "error" not in sus and sus["seconds"] >= 0.95 and 0.85 < sus["ratio_to_headline_value"] < 1.15 and 0.3 < sus["roofline_frac"] < 0.66
    assert rec["two_streams"]["steps_run"] >= 20 and rec["two_streams"]["ratio_to_headline_value"] > 0.9
    assert rec["roofline"]["kernel_ms_avg_per_launch_pass"] < 1.1 * rec["roofline"]["kernel_ms_avg"]    # (round 5: 1.22 -- a cold clock)
    assert rec["config"]["versions"]["rocm"] and rec["config"]["versions"]["hip_runtime"]
    cfgs = rec["configs"]
    assert set(cfgs) == {"cfg3", "cfg5", "chunk_l1", "cfg4_band"}, cfgs.get("error")
    for name, c in cfgs.items():
        assert "error" not in c, (name, c)
        assert c["output_verified"]["verified"] is True, (name, c["output_verified"])
        assert c["value"] > 1000.0 and 0.2 < c["roofline"]["frac"] < 0.66, (name, c["value"], c["roofline"]["frac"])
        assert c["steps_run"] >= c["steps"] and c["steps_run"] * c["roofline"]["kernel_ms_avg"] >= 45.0, (name, c["steps_run"])
    assert cfgs["cfg4_band"]["dtype"] == "f32" and cfgs["cfg3"]["pixel_iterations_per_step"] == 65146485486
    e2e = rec["end_to_end"]
    assert e2e["slots"] == 4 and e2e["tiles_per_s_all_slots_in_flight_lazy_uniform"] > e2e["tiles_per_s_synchronous"]
    assert e2e["worker_depth"] == 3 and e2e["tiles_per_s_three_in_flight_lazy_uniform"] > e2e["tiles_per_s_synchronous"]
    assert e2e["host_answered_tiles"] == 32      # level 16: the corners of [-2, 2]^2 outside |c| = 2

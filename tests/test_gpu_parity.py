"""GPU parity tests (run on the MI355X box with -m gpu).  Everything goes through the C ABI
(libmbk_hip.so via ctypes); the CPU oracle and the committed golden vectors are the checkers.
The bar is bit-exact: int32 escape indices and uint8 quantised bytes."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from distributedmandelbrot_amd import View
from distributedmandelbrot_amd import _lib as L

pytestmark = pytest.mark.gpu

KERNELS = ["default", "simple", "asm", "refill", "group", "scan"]


def _check_view(gpu, oracle, view, mrd, window=None, kernel="default"):
    c, b, st = gpu.compute_view(view, mrd, window=window, kernel=kernel)
    oc, ob, total = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, view.width,
                                view.height, mrd, window=window)
    assert np.array_equal(c, oc), (view, mrd, window, kernel, int((c != oc).sum()))
    assert np.array_equal(b, ob), (view, mrd, window, kernel)
    assert st.pixel_iterations == total
    assert st.never_pixels == int((oc == 0).sum())
    assert st.all_bytes_zero == bool((ob == 0).all()) and st.all_bytes_one == bool((ob == 1).all())
    return c, b, st


@pytest.mark.parametrize("kernel", KERNELS)
def test_golden_small_windows(gpu, golden, kernel):
    """Vectors produced by the reference's own gen_arrays + calc_mb_value (make_golden.py)."""
    for name in golden["small/names"]:
        sr, si, rng, n, mrd = golden[f"small/{name}/params"]
        n, mrd = int(n), int(mrd)
        c, _, _ = gpu.compute_view(View(sr, si, rng, rng, n, n), mrd, want_bytes=False, kernel=kernel)
        assert np.array_equal(c, golden[f"small/{name}/counts"]), (name, kernel)


@pytest.mark.parametrize("kernel", KERNELS)
def test_golden_points(gpu, golden, kernel):
    """Known-answer points, each as a 1x1 view (x[0] = start)."""
    for (cr, ci, mrd), ref in zip(golden["points/inputs"], golden["points/counts"]):
        c, _, _ = gpu.compute_view(View(cr, ci, 1.0, 1.0, 1, 1), int(mrd), want_bytes=False, kernel=kernel)
        assert int(c[0, 0]) == int(ref), (cr, ci, mrd, kernel)


@pytest.mark.parametrize("kernel", KERNELS)
def test_golden_full_datachunks(gpu, golden, kernel):
    """Full 4096x4096 tiles from the reference's unmodified process_workload, via mbk_datachunk."""
    for key in golden["full/names"]:
        level, mrd, ir, ii = (int(x) for x in golden[f"full/{key}/params"])
        if kernel == "default":
            byts, counts, st = gpu.datachunk(level, mrd, ir, ii, want_counts=True)
        else:
            sr, si, rng = __import__("distributedmandelbrot_amd.device", fromlist=["x"]).datachunk_geometry(level, ir, ii)
            counts, byts, st = gpu.compute_view(View(sr, si, rng, rng, 4096, 4096), mrd, kernel=kernel)
        assert hashlib.sha256(byts.tobytes()).hexdigest() == str(golden[f"full/{key}/bytes_sha256"]), (key, kernel)
        assert hashlib.sha256(counts.astype("<i4").tobytes()).hexdigest() == \
            str(golden[f"full/{key}/counts_sha256"]), (key, kernel)
        assert st.never_pixels == int(golden[f"full/{key}/zeros"])
        assert st.all_bytes_zero == (int(golden[f"full/{key}/zeros"]) == 16777216)   # (20,1024,7,9) is a Never chunk


@pytest.mark.parametrize("kernel", KERNELS)
def test_seeded_views_against_oracle(gpu, oracle, kernel):
    rs = np.random.RandomState(20260921)
    cases = [
        (View(-2.0, -1.5, 3.0, 3.0, 512, 512), 256),           # BASELINE cfg1
        (View(-0.743648, 0.131820, 1e-5, 1e-5, 192, 160), 10000),  # cfg3 window, deep zoom
        (View(-0.755, 0.10, 0.02, 0.02, 300, 200), 1024),      # seahorse valley
        (View(-2.0, -2.0, 4.0, 4.0, 257, 129), 300),
    ]
    for _ in range(6):
        cr, ci = rs.uniform(-1.6, 0.4), rs.uniform(-1.1, 1.1)
        span = 10.0 ** rs.uniform(-7, 0)
        cases.append((View(cr, ci, span, span * rs.uniform(0.5, 2.0), int(rs.randint(1, 400)),
                           int(rs.randint(1, 400))), int(rs.randint(2, 1500))))
    for view, mrd in cases:
        _check_view(gpu, oracle, view, mrd, kernel=kernel)


@pytest.mark.parametrize("kernel", KERNELS)
def test_ragged_shapes_and_windows(gpu, oracle, kernel):
    """Sizes that are not multiples of the 8x8 / 32x8 blocks, 1-wide/1-high views, offset windows."""
    base = View(-1.3, -0.4, 1.1, 0.9, 77, 53)
    for w, h in [(1, 1), (1, 64), (64, 1), (7, 9), (8, 8), (9, 8), (31, 33), (32, 8), (33, 9), (65, 17)]:
        _check_view(gpu, oracle, View(-0.9, 0.05, 0.6, 0.45, w, h), 200, kernel=kernel)
    for window in [(0, 0, 77, 53), (5, 7, 40, 30), (76, 52, 1, 1), (0, 10, 77, 3), (13, 0, 1, 53)]:
        _check_view(gpu, oracle, base, 333, window=window, kernel=kernel)


@pytest.mark.parametrize("kernel", KERNELS)
def test_mrd_edge_cases(gpu, oracle, kernel):
    v = View(-2.0, -1.25, 2.5, 2.5, 64, 48)
    for mrd in (1, 2, 3):
        _check_view(gpu, oracle, v, mrd, kernel=kernel)
    c, _, st = gpu.compute_view(v, 0, want_bytes=False, kernel=kernel)   # range(1, 0) is empty
    assert not c.any() and st.pixel_iterations == 0
    # 64-bit quantiser path (count*256 + mrd - 1 >= 2^32): exterior-only window keeps it cheap
    far = View(1.5, 1.5, 0.5, 0.5, 40, 40)
    _check_view(gpu, oracle, far, 2 ** 31 - 1, kernel=kernel)
    _check_view(gpu, oracle, far, 2 ** 24 + 3, kernel=kernel)


def test_argument_errors(gpu):
    from distributedmandelbrot_amd import MbkError
    v = View(-2.0, -2.0, 4.0, 4.0, 16, 16)
    with pytest.raises(MbkError):
        gpu.compute_view(v, 0, want_bytes=True)                       # quantiser would divide by zero
    with pytest.raises(MbkError):
        gpu.compute_view(v, 2 ** 31, want_bytes=False)                # int32 result type
    with pytest.raises(MbkError):
        gpu.compute_view(v, 10, window=(10, 0, 7, 16))                # window outside the view
    with pytest.raises(MbkError):
        gpu._check(gpu._lib.mbk_view_compute(gpu._h, C.byref(L.mbk_view(-2.0, -2.0, 4.0, 4.0, 16, 16, 0, 0, 0, 4)),
                                             10, L.MBK_WANT_COUNTS, np.empty(4, np.int32).ctypes.data, None, None))  # empty window
    with pytest.raises(MbkError):
        gpu._check(gpu._lib.mbk_view_compute(gpu._h, C.byref(L.mbk_view(-2.0, -2.0, 4.0, 4.0, 0, 16, 0, 0, 1, 1)),
                                             10, L.MBK_WANT_COUNTS, np.empty(4, np.int32).ctypes.data, None, None))  # empty view
    with pytest.raises(MbkError):
        gpu.compute_view(View(float("nan"), 0.0, 1.0, 1.0, 4, 4), 10)
    with pytest.raises(MbkError):
        gpu.compute_view(View(0.0, 0.0, float("inf"), 1.0, 4, 4), 10)
    with pytest.raises(MbkError):
        gpu.datachunk(4, 256, 4, 0)                                   # index >= level (DataChunk.cs:102-106)
    with pytest.raises(KeyError):
        gpu.compute_view(v, 10, kernel="nonexistent")


def test_tiny_imaginary_coordinates_use_exact_doubling(gpu, oracle):
    """fma(2, zr*zi, ci) differs from fl(fl((2 zr) zi) + ci) only when zr*zi is subnormal and ci tiny;
    the library must detect such views and stay bit-exact."""
    for start_i, range_i in [(1e-310, 3e-310), (-4e-320, 9e-320), (0.0, 1e-305), (2e-300, 1e-301)]:
        _check_view(gpu, oracle, View(-1.8, start_i, 2.2, range_i, 96, 24), 500)
    _check_view(gpu, oracle, View(-1.9, -1e-308, 0.4, 2e-308, 200, 9), 3000)


def test_step_zero_linspace_fallback(gpu, oracle):
    _check_view(gpu, oracle, View(-0.75, 0.1, 0.0, 0.0, 5, 4), 100)
    _check_view(gpu, oracle, View(0.3, 0.0, 1e-3, 5e-324, 9, 6), 100)


@pytest.mark.parametrize("kernel", KERNELS)
def test_full_size_cfg2_bit_exact_and_band_invariant(gpu, oracle, kernel):
    """BASELINE cfg2 at full size (4096^2, mrd 1000): bit-exact against the oracle run on the host
    cores, and a banded evaluation (the multi-GPU shard unit) equals the whole."""
    view, mrd = View(-2.0, -1.5, 3.0, 3.0, 4096, 4096), 1000
    c, b, st = gpu.compute_view(view, mrd, kernel=kernel)
    oc, ob, total = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, 4096, 4096, mrd)
    assert np.array_equal(c, oc) and np.array_equal(b, ob)
    assert st.pixel_iterations == total == int(np.where(c > 0, c, mrd - 1).astype(np.int64).sum())
    for row0, nrows in [(0, 128), (1024, 100), (4000, 96)]:
        cb, bb, _ = gpu.compute_view(view, mrd, window=(0, row0, 4096, nrows), kernel=kernel)
        assert np.array_equal(cb, c[row0:row0 + nrows]) and np.array_equal(bb, b[row0:row0 + nrows])


def test_device_pointer_launch_and_reduce_via_torch(gpu):
    """mbk_view_launch on torch-owned HBM on torch's current stream (what bench.py times)."""
    import torch
    view, mrd = View(-2.0, -1.5, 3.0, 3.0, 1024, 768), 500
    d_counts = torch.empty(768 * 1024, dtype=torch.int32, device="cuda:0")
    d_bytes = torch.empty(768 * 1024, dtype=torch.uint8, device="cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    gpu.launch_view(view, mrd, d_counts=d_counts.data_ptr(), d_bytes=d_bytes.data_ptr(), stream=s)
    st = gpu.reduce_counts(d_counts.data_ptr(), d_counts.numel(), mrd, stream=s)
    torch.cuda.synchronize()
    c, b, st2 = gpu.compute_view(view, mrd)
    assert np.array_equal(d_counts.cpu().numpy().reshape(768, 1024), c)
    assert np.array_equal(d_bytes.cpu().numpy().reshape(768, 1024), b)
    assert st.pixel_iterations == st2.pixel_iterations and st.never_pixels == st2.never_pixels


def test_worker_process_workload_on_gpu(golden):
    """The reference's in-process seam: process_workload(level, mrd, ir, ii) -> uint8[16777216]."""
    from distributedmandelbrot_amd import worker
    key = "4_256_0_0"
    out = worker.process_workload(4, 256, 0, 0)
    assert out.dtype == np.uint8 and out.shape == (16777216,)
    assert hashlib.sha256(out.tobytes()).hexdigest() == str(golden[f"full/{key}/bytes_sha256"])


def test_worker_end_to_end_against_fake_distributer(golden):
    """Distributer protocol + HIP compute, two tiles, checked against the reference-made goldens."""
    from distributedmandelbrot_amd import worker
    from fake_distributer import FakeDistributer
    with FakeDistributer([(4, 256)]) as srv:
        # lease tiles (4,256,0,0), (4,256,0,1), (4,256,0,2) ... ; compute the first three
        for _ in range(3):
            assert worker.do_workload_single("127.0.0.1", srv.port, log=lambda *a: None)
        assert srv.wait_completed(3)
        data = srv.completed[(4, 256, 0, 0)]
        assert hashlib.sha256(data.tobytes()).hexdigest() == str(golden["full/4_256_0_0/bytes_sha256"])


def test_render_view_multi_queue_single_gpu(gpu, oracle):
    """The per-GPU work queue with one real device: banded result equals the oracle."""
    from distributedmandelbrot_amd.sharding import render_view
    view, mrd = View(-0.755, 0.10, 0.02, 0.02, 640, 500), 800
    c, b, per = render_view([gpu], view, mrd, band_rows=64)
    oc, ob, total = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, 640, 500, mrd)
    assert np.array_equal(c, oc) and np.array_equal(b, ob)
    assert per[0]["bands"] == 8 and per[0]["pixel_iterations"] == total


@pytest.mark.parametrize("kernel", ["default", "asm", "group", "scan"])
def test_f32_variant_against_f32_oracle(gpu, oracle, kernel):
    """BASELINE cfg4 (fp32 kernel variant): bit-exact against the strict-binary32 oracle."""
    rs = np.random.RandomState(4)
    cases = [
        (View(-2.0, -1.5, 3.0, 3.0, 512, 384), 256),
        (View(-0.755, 0.10, 0.02, 0.02, 256, 256), 50000),     # cfg4's region and mrd, small window
        (View(-2.0, -2.0, 4.0, 4.0, 130, 70), 300),             # crosses the |c| = 2 ring (per-step path)
        (View(-0.75, -1e-40, 0.5, 2e-40, 64, 8), 500),         # tiny imaginary parts -> exact doubling
        (View(1.0e3, 1.0e3, 1.0, 1.0, 9, 9), 50),
    ]
    for _ in range(4):
        cr, ci = rs.uniform(-1.6, 0.4), rs.uniform(-1.1, 1.1)
        span = 10.0 ** rs.uniform(-4, 0)
        cases.append((View(cr, ci, span, span, int(rs.randint(1, 300)), int(rs.randint(1, 300))), int(rs.randint(2, 1500))))
    for view, mrd in cases:
        c, b, st = gpu.compute_view(view, mrd, kernel=kernel, precision="f32")
        oc, ob, total = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, view.width,
                                    view.height, mrd, precision="f32")
        assert np.array_equal(c, oc), (view, mrd, kernel, int((c != oc).sum()))
        assert np.array_equal(b, ob) and st.pixel_iterations == total
    from distributedmandelbrot_amd import MbkError
    with pytest.raises(MbkError):
        gpu.compute_view(View(-2.0, -1.5, 3.0, 3.0, 16, 16), 10, kernel="simple", precision="f32")
    with pytest.raises(MbkError):
        gpu.compute_view(View(1e30, 0.0, 1.0, 1.0, 4, 4), 10, precision="f32")   # beyond the fp32 domain


@pytest.mark.parametrize("kernel", ["default", "asm", "group", "scan"])
def test_smooth_colouring_cfg5(gpu, oracle, kernel):
    """BASELINE cfg5: integer part (the count) bit-exact, the continuous value within 1e-12 of the libm
    evaluation of the same formula on the same |z_n|^2."""
    cases = [(View(-2.0, -1.5, 3.0, 3.0, 512, 512), 5000), (View(-0.755, 0.10, 0.02, 0.02, 300, 200), 5000),
             (View(-2.0, -2.0, 4.0, 4.0, 65, 33), 40), (View(-0.1, -0.1, 0.2, 0.2, 16, 16), 100)]
    for view, mrd in cases:
        sm, c, st = gpu.compute_view_smooth(view, mrd, kernel=kernel)
        osm, oc = oracle.view_smooth(view.start_r, view.start_i, view.range_r, view.range_i, view.width, view.height, mrd)
        assert np.array_equal(c, oc)
        esc = oc > 0
        assert (sm[~esc] == 0.0).all()
        assert np.allclose(sm[esc], osm[esc], rtol=0, atol=1e-12 * max(1, mrd)), float(np.abs(sm[esc] - osm[esc]).max())
        # nu lies in (n, n + 1 - log2(0.5 ln 4)] because |z_n|^2 >= 4
        assert (sm[esc] <= oc[esc] + 1.0 - np.log2(0.5 * np.log(4.0)) + 1e-12).all()


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_fuzz_many_small_views(gpu, oracle, precision):
    """120 seeded random small views (random centre incl. the |c| = 2 ring and the set boundary, random
    aspect, window and mrd): default kernel bit-exact against the oracle of the same precision."""
    rs = np.random.RandomState(77 if precision == "f64" else 78)
    for k in range(120):
        kind = k % 4
        if kind == 0:      # anywhere in the reference's domain [-2,2]^2
            cr, ci = rs.uniform(-2, 2), rs.uniform(-2, 2)
        elif kind == 1:    # on the |c| = 2 circle (grouped test must fall back to per-step)
            th = rs.uniform(0, 2 * np.pi)
            cr, ci = 2 * np.cos(th), 2 * np.sin(th)
        elif kind == 2:    # near the boundary of the main cardioid
            th = rs.uniform(0, 2 * np.pi)
            cr, ci = 0.5 * np.cos(th) - 0.25 * np.cos(2 * th), 0.5 * np.sin(th) - 0.25 * np.sin(2 * th)
        else:              # real axis / antenna
            cr, ci = rs.uniform(-2, 0.3), 0.0
        span_r = 10.0 ** rs.uniform(-9, 0.3)
        span_i = span_r * rs.uniform(0.3, 3.0)
        w, h = int(rs.randint(1, 48)), int(rs.randint(1, 48))
        mrd = int(rs.choice([2, 3, 9, 10, 17, 18, 33, 100, 257, 700]))
        view = View(cr - span_r / 2, ci - span_i / 2, span_r, span_i, w, h)
        window = None
        if w > 4 and h > 4 and k % 3 == 0:
            c0, r0 = int(rs.randint(0, w - 2)), int(rs.randint(0, h - 2))
            window = (c0, r0, int(rs.randint(1, w - c0 + 1)), int(rs.randint(1, h - r0 + 1)))
        c, b, st = gpu.compute_view(view, mrd, window=window, precision=precision)
        oc, ob, total = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, w, h, mrd,
                                    window=window, precision=precision)
        assert np.array_equal(c, oc), (k, view, mrd, window, int((c != oc).sum()))
        assert np.array_equal(b, ob) and st.pixel_iterations == total, (k, view, mrd)


def test_two_tiles_in_flight_submit_wait(gpu, golden):
    """mbk_datachunk_submit / mbk_wait: two slots, results identical to the synchronous path (goldens)."""
    from distributedmandelbrot_amd import MbkError
    bufs = [gpu.pinned_empty((16777216,), np.uint8) for _ in range(2)]
    keys = ["4_256_0_0", "10_1024_0_5", "4_256_1_2", "4_256_0_0"]
    params = [tuple(int(x) for x in golden[f"full/{k}/params"]) for k in keys]
    gpu.submit_datachunk(0, *params[0], bufs[0])
    with pytest.raises(MbkError):
        gpu.submit_datachunk(0, *params[1], bufs[0])              # slot busy
    for i in range(1, len(params) + 1):                           # software pipeline: submit i, wait i-1
        if i < len(params):
            gpu.submit_datachunk(i % 2, *params[i], bufs[i % 2])
        st = gpu.wait((i - 1) % 2)
        got = hashlib.sha256(bufs[(i - 1) % 2].tobytes()).hexdigest()
        assert got == str(golden[f"full/{keys[i - 1]}/bytes_sha256"]), keys[i - 1]
        assert st.never_pixels == int(golden[f"full/{keys[i - 1]}/zeros"])
    with pytest.raises(MbkError):
        gpu.wait(0)                                               # nothing in flight


# ------------------------------------------------------------------------------------------------------
# Round 2: BASELINE configs at their NAMED sizes, the option matrix, and regression tests for round-1 bugs
# ------------------------------------------------------------------------------------------------------

CFG3 = (View(-0.743648, 0.131820, 1e-5, 1e-5, 8192, 8192), 10000)      # BASELINE configs[2]
CFG4 = (View(-0.755, 0.10, 0.02, 0.02, 16384, 16384), 50000)           # BASELINE configs[3] (fp32)
CFG5 = (View(-2.0, -1.5, 3.0, 3.0, 4096, 4096), 5000)                  # BASELINE configs[4] (smooth)


@pytest.fixture(scope="module")
def cfg3_oracle_counts(oracle):
    """cfg3 at full size from the 8-lane AVX-512 evaluation of the oracle (bit-identical to the scalar
    one: tests/test_oracle.py); ~65 G pixel-iterations, a few seconds on the GPU box's host cores."""
    view, mrd = CFG3
    if not oracle.have_avx512():
        pytest.skip("host without AVX-512: the scalar oracle needs minutes for cfg3")
    counts, total = oracle.view_avx512(view.start_r, view.start_i, view.range_r, view.range_i, view.width,
                                       view.height, mrd)
    return counts, total


@pytest.mark.parametrize("kernel", ["default", "refill", "group"])
def test_full_size_cfg3_deep_zoom(gpu, cfg3_oracle_counts, kernel):
    """BASELINE cfg3 (8192^2 deep zoom, mrd 10000) at its named size: counts bit-exact, bytes equal to the
    quantised oracle counts; `refill` (persistent lane-refill kernel + edge strips) included."""
    view, mrd = CFG3
    oc, total = cfg3_oracle_counts
    c, b, st = gpu.compute_view(view, mrd, kernel=kernel)
    assert np.array_equal(c, oc), (kernel, int((c != oc).sum()))
    assert st.pixel_iterations == total and st.never_pixels == int((oc == 0).sum())
    ob = ((oc.astype(np.int64) * 256 + mrd - 1) // mrd).astype(np.uint8)     # SURVEY.md A.4 integer form
    assert np.array_equal(b, ob), kernel
    # row bands (the multi-GPU shard unit) at full width equal the whole
    for row0, nrows in [(0, 128), (4096 - 64, 128), (8192 - 72, 72)]:
        cb, _, _ = gpu.compute_view(view, mrd, window=(0, row0, 8192, nrows), want_bytes=False, kernel=kernel)
        assert np.array_equal(cb, oc[row0:row0 + nrows]), (kernel, row0)


@pytest.mark.parametrize("kernel", ["default", "group"])
def test_cfg4_full_width_bands_f32(gpu, oracle, kernel):
    """BASELINE cfg4 (16384^2 seahorse valley, mrd 50000, fp32 variant): 8 full-width 8-row bands spread
    over the image, bit-exact against the strict-binary32 oracle."""
    view, mrd = CFG4
    for row0 in [0, 2048 + 8, 4096, 6000, 8192 - 8, 10240, 13000, 16384 - 8]:
        window = (0, row0, 16384, 8)
        c, b, st = gpu.compute_view(view, mrd, window=window, kernel=kernel, precision="f32")
        oc, ob, total = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, view.width,
                                    view.height, mrd, window=window, precision="f32")
        assert np.array_equal(c, oc), (kernel, row0, int((c != oc).sum()))
        assert np.array_equal(b, ob) and st.pixel_iterations == total, (kernel, row0)


def test_cfg4_full_image_two_independent_loops_agree_f32(gpu):
    """BASELINE cfg4 at its FULL size (16384^2, mrd 50000, fp32): the CPU oracle would need hours for the whole image, so
    the whole-image check is a size-independent one.  The library's default path (grouped bailout with the deferred replay,
    heavy-first dispatch list, cycle test on: iterations skipped) and kernel "asm" with the cycle test off (one exact test per
    step, an escaped lane leaves EXEC, every iteration executed, image order) are two separately written loops; they must
    agree on all 268 435 456 counts, and the 1 024-row band of the image that the strict-binary32 oracle DID compute
    (tests/golden/bench_outputs.json "cfg4_band": 115 s on the CPU) must hash to the oracle's sha256 inside it."""
    import hashlib
    import json
    import os
    import torch
    from distributedmandelbrot_amd import MandelbrotDevice
    view, mrd = CFG4
    n = view.width * view.height
    s = torch.cuda.current_stream().cuda_stream
    a = torch.full((n,), -7, dtype=torch.int32, device="cuda:0")
    gpu.launch_view(view, mrd, d_counts=a.data_ptr(), stream=s, precision="f32")
    st_a = gpu.reduce_counts(a.data_ptr(), n, mrd, stream=s)
    b = torch.full((n,), -9, dtype=torch.int32, device="cuda:0")
    with MandelbrotDevice(0) as strict:
        strict.set_option("cycle_detect", 0)
        strict.launch_view(view, mrd, d_counts=b.data_ptr(), stream=s, precision="f32", kernel="asm")
        st_b = strict.reduce_counts(b.data_ptr(), n, mrd, stream=s)
        torch.cuda.synchronize()
    assert torch.equal(a, b), int((a != b).sum())
    assert st_a.pixel_iterations == st_b.pixel_iterations and st_a.never_pixels == st_b.never_pixels
    assert int(a.min()) >= 0 and int(a.max()) <= mrd - 1
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bench_outputs.json")))["cfg4_band"]
    assert g["view"] == [view.start_r, view.start_i, view.range_r, view.range_i, view.width, view.height] and g["mrd"] == mrd
    _, row0, ncols, nrows = g["window"]
    band = a.view(view.height, view.width)[row0:row0 + nrows].contiguous().cpu().numpy()
    assert ncols == view.width and hashlib.sha256(band.tobytes()).hexdigest() == g["counts_sha256"]
    assert int(np.where(band > 0, band, mrd - 1).astype(np.int64).sum()) == g["pixel_iterations"]
    assert int((band == 0).sum()) == g["never_pixels"]


def test_full_size_cfg5_smooth(gpu, oracle):
    """BASELINE cfg5 (4096^2, mrd 5000, continuous colouring) at its named size: counts bit-exact, the
    continuous value within 1e-12 * mrd of the libm evaluation on the same |z_n|^2."""
    view, mrd = CFG5
    sm, c, st = gpu.compute_view_smooth(view, mrd)
    osm, oc = oracle.view_smooth(view.start_r, view.start_i, view.range_r, view.range_i, view.width, view.height, mrd)
    assert np.array_equal(c, oc), int((c != oc).sum())
    esc = oc > 0
    assert (sm[~esc] == 0.0).all()
    assert float(np.abs(sm[esc] - osm[esc]).max()) <= 1e-12 * mrd
    assert st.pixel_iterations == int(np.where(oc > 0, oc, mrd - 1).astype(np.int64).sum())


def test_smooth_buffer_survives_serialize_last(gpu, oracle):
    """Round-1 bug: mbk_serialize_last freed the smooth buffer while growing its RLE scratch; the next
    compute_view_smooth wrote into freed HBM.  Sequence smooth -> datachunk -> serialize_last -> smooth on a
    FRESH ctx (so that the RLE scratch really grows after the smooth buffer exists)."""
    from distributedmandelbrot_amd import MandelbrotDevice
    from oracle.serializer import serialize
    view, mrd = View(-0.755, 0.10, 0.02, 0.02, 700, 500), 900
    osm, oc = oracle.view_smooth(view.start_r, view.start_i, view.range_r, view.range_i, view.width, view.height, mrd)
    with MandelbrotDevice(0) as dev:
        sm1, c1, _ = dev.compute_view_smooth(view, mrd)
        byts, _, _ = dev.datachunk(10, 1024, 3, 5)
        stream, codec = dev.serialize_last()
        assert stream == serialize(byts)
        sm2, c2, _ = dev.compute_view_smooth(view, mrd)
        byts2, _, _ = dev.datachunk(10, 1024, 3, 5)          # device buffers still intact afterwards
        sm3, c3, _ = dev.compute_view_smooth(view, mrd)
    for sm, c in ((sm1, c1), (sm2, c2), (sm3, c3)):
        assert np.array_equal(c, oc)
        assert np.allclose(sm[oc > 0], osm[oc > 0], rtol=0, atol=1e-12 * mrd) and (sm[oc == 0] == 0).all()
    assert np.array_equal(byts, byts2)


OPTION_MATRIX = [
    ("scan", {"exact_steps": 0}), ("scan", {"exact_steps": 3}),
    ("scan", {"exact_steps": 64}), ("scan", {"scan_waves": 1}), ("scan", {"scan_xcd_map": 0}), ("scan", {"scan_xcd_map": 1}),
    ("scan", {"scan_col_period": 0}), ("scan", {"scan_col_period": 1, "scan_xcd_map": 0}), ("scan", {"scan_waves": 3, "exact_steps": 16}),
    ("group", {"exact_steps": 0}), ("group", {"exact_steps": 5}), ("group", {"group_steps": 4}), ("group", {"group_steps": 8}), ("group", {"group_steps": 16, "order": 0}),
    ("scan", {"group_steps": 8}),
    ("group", {"waves_per_wg": 2}), ("group", {"waves_per_wg": 4, "order": 1}), ("group", {"order": 0}),
    ("group", {"order": 1}), ("group", {"probe_steps": 2}), ("asm", {"waves_per_wg": 4, "order": 0}),
    ("refill", {"rf_livemin": 0}), ("refill", {"rf_livemin": 63, "rf_patience": 16}),
    ("refill", {"rf_batch": 4, "rf_waves": 2}),
    ("group", {"cycle_detect": 0}), ("scan", {"cycle_detect": 0}), ("default", {"cycle_detect": 0}),
    ("group", {"cycle_detect": 0, "group_steps": 8, "waves_per_wg": 2}), ("group", {"cycle_detect": 1, "exact_steps": 0, "order": 0}),
    ("group", {"probe_mid": 2}), ("group", {"probe_mid": 65537, "cycle_detect": 0}), ("default", {"probe_mid": 20, "probe_steps": 64}),
    ("default", {"heavy_share": 0}), ("default", {"heavy_share": 65536}),
    ("group", {"prepass_overlap": 0}), ("default", {"prepass_overlap": 0, "probe_mid": 6}),
    ("group", {"exact_long": 0}), ("scan", {"exact_long": 0, "cycle_detect": 0}), ("default", {"exact_long": 3, "exact_steps": 5}),
    ("group", {"group_steps": 32, "cycle_detect": 0}), ("group", {"group_steps": 32}), ("default", {"group_steps": 32, "cycle_detect": 0, "exact_long": 8}),
    ("scan", {"group_steps": 32, "cycle_detect": 0}),
    ("scan", {"scan_inline": 0}), ("default", {"scan_inline": 0, "cycle_detect": 0}), ("scan", {"scan_inline": 1, "scan_waves": 2, "cycle_detect": 0}),
    ("group", {"wave_limit": 4}), ("scan", {"wave_limit": 2, "scan_inline": 0}), ("default", {"wave_limit": 7, "cycle_detect": 0}),
    ("group", {"order": 3, "units_min_light": 0}), ("group", {"order": 3, "cycle_detect": 0, "units_min_light": 0}),
    ("default", {"order": 3, "exact_steps": 3, "probe_steps": 8, "units_min_light": 0}), ("group", {"order": 3, "units_min_light": 65536}),
    ("group", {"order": 2}), ("group", {"order": 3, "group_steps": 8}), ("group", {"order": 3, "waves_per_wg": 2}),
    ("group", {"order": 3, "xcd_balance": 0}), ("group", {"order": 3, "xcd_balance": 2, "units_min_light": 0}),
    ("default", {"xcd_balance": 2, "cycle_detect": 0}),
    ("group", {"order": 3, "m_late": 0}), ("group", {"order": 3, "m_late": 4, "cycle_detect": 0, "xcd_balance": 2, "units_min_light": 0}),
    ("default", {"m_late": 31}), ("group", {"order": 3, "m_late": 12, "units_min_light": 0}), ("default", {"m_late": 65536, "xcd_balance": 1, "cycle_detect": 0}),
    ("group", {"order": 3, "h_settled": 0}), ("group", {"order": 3, "h_settled": 1, "m_late": 0, "xcd_balance": 2, "units_min_light": 0}),
    ("default", {"h_settled": 30}), ("group", {"order": 3, "h_settled": 9, "cycle_detect": 0}),
    ("default", {"prepass_overlap": 2, "classify_wg": 256}), ("group", {"classify_wg": 64, "m_late": 4}),
    ("scan", {"scan_strip": 0}), ("default", {"scan_strip": 0, "cycle_detect": 0, "scan_waves": 3}),
    ("group", {"cycle_window": 0}), ("default", {"cycle_window": 5}), ("scan", {"cycle_window": 65536, "group_steps": 8}),
    ("default", {"cycle_window": 1, "h_settled": 0, "m_late": 0}),
    ("group", {"spill_first": 32, "spill_lanes": 32, "spill_min_mrd": 2, "spill_min_blocks": 0, "order": 2}), ("default", {"spill_first": 0}),
    ("default", {"spill_first": 64, "spill_lanes": 3, "spill_min_mrd": 100, "spill_min_blocks": 3, "units_min_light": 65536, "cycle_detect": 0}),
]


@pytest.mark.parametrize("kernel,options", OPTION_MATRIX, ids=lambda x: str(x))
def test_option_matrix_is_bit_exact(oracle, kernel, options):
    """Every tuning option the library accepts (mbk_set_option; no environment variable is read) changes
    scheduling only.  A fresh ctx per case; seeded small views + cfg2 at full size against the oracle."""
    from distributedmandelbrot_amd import MandelbrotDevice, MbkError
    rs = np.random.RandomState(5)
    cases = [(View(-2.0, -1.5, 3.0, 3.0, 4096, 4096), 1000), (View(-2.0, -2.0, 4.0, 4.0, 300, 200), 40),
             (View(-0.743648, 0.131820, 1e-5, 1e-5, 200, 136), 4000), (View(-0.2, -0.1, 0.2, 0.2, 64, 64), 100)]
    for _ in range(10):
        cr, ci = rs.uniform(-1.6, 0.4), rs.uniform(-1.1, 1.1)
        span = 10.0 ** rs.uniform(-6, 0)
        cases.append((View(cr, ci, span, span * rs.uniform(0.5, 2.0), int(rs.randint(1, 300)), int(rs.randint(1, 300))),
                      int(rs.choice([2, 9, 10, 17, 18, 24, 25, 26, 33, 41, 100, 700]))))
    with MandelbrotDevice(0) as dev:
        for name, value in options.items():
            dev.set_option(name, value)
            assert dev.get_option(name) == value
        with pytest.raises(MbkError):
            dev.set_option("group_steps", 5)
        with pytest.raises(MbkError):
            dev.set_option("scan_waves", 9)
        for view, mrd in cases:
            _check_view(dev, oracle, view, mrd, kernel=kernel)


SPILL_OPTIONS = [{"spill_first": 256, "spill_lanes": 16}, {"spill_first": 32, "spill_lanes": 1}, {"spill_first": 32, "spill_lanes": 32},
                 {"spill_first": 64, "spill_lanes": 5, "exact_steps": 0}, {"spill_first": 512, "spill_lanes": 8, "exact_steps": 21, "exact_long": 3},
                 {"spill_first": 96, "spill_lanes": 16, "cycle_window": 0, "spill_cyc_shift": 31}, {"spill_first": 64, "spill_lanes": 9, "spill_cyc_shift": 0}, {"spill_first": 128, "spill_lanes": 16, "order": 2, "prepass_overlap": 0}]


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("cycle", [0, 1])
def test_spill_second_pass_is_bit_exact(oracle, precision, cycle):
    """SPILL (round 6, MBK_OPT_SPILL_FIRST / _LANES / _MIN_MRD; csrc/mbk_kernels.h block_pixel_spill, csrc/mbk_spill.h): blocks
    that reach a checkpoint with a few live lanes hand them to a second pass.  Deep-zoom windows (filaments: where it pays),
    a ragged window, a window with an offset, a full-set view forced onto the path (ring waves, blocks that are all set) --
    every output set, checkpoints from 32 steps on, 1 .. 32 lanes, against the oracle; and the second pass really ran."""
    from distributedmandelbrot_amd import MandelbrotDevice
    views = [(View(-0.743648, 0.131820, 1e-5, 1e-5, 1024, 1024), 3000, None, "default", {}),
             (View(-0.7436431, 0.1318255, 2e-7, 2e-7, 1200, 1100), 2500, (3, 5, 1101, 1027), "default", {}),      # ragged, offset
             (View(-0.755, 0.10, 0.02, 0.02, 1024, 1024), 2048, None, "group", {}),
             (View(-2.0, -1.5, 3.0, 3.0, 1024, 1024), 700, None, "group", {"order": 2})]                        # the |c| = 2 ring, interior blocks
    if precision == "f32":   # (fp32 has no resolution at 1e-7: the deep windows become blocky, which the oracle reproduces)
        views = [views[0], views[2], views[3]]
    want = {}
    for i, (view, mrd, window, kernel, extra) in enumerate(views):
        want[i] = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, view.width, view.height, mrd, window=window, precision=precision)
    for k, opts in enumerate(SPILL_OPTIONS):
        with MandelbrotDevice(0) as dev:
            dev.set_option("cycle_detect", cycle)
            dev.set_option("spill_min_mrd", 2)
            dev.set_option("spill_min_blocks", 0)
            for name, value in opts.items():
                dev.set_option(name, value)
            for i, (view, mrd, window, kernel, extra) in enumerate(views):
                for name, value in extra.items():
                    dev.set_option(name, value)
                before = dev.spill_info()["launches"]
                wc, wb = (True, True) if (i + k) % 3 == 0 else (True, False) if (i + k) % 3 == 1 else (False, True)
                c, b, st = dev.compute_view(view, mrd, window=window, kernel=kernel, precision=precision, want_counts=wc, want_bytes=wb)
                oc, ob, total = want[i]
                if wc:
                    assert np.array_equal(c, oc), (opts, i, int((c != oc).sum()))
                if wb:
                    assert np.array_equal(b, ob), (opts, i)
                assert st.pixel_iterations == total and st.never_pixels == int((oc == 0).sum())
                info = dev.spill_info()
                assert info["launches"] == before + 1, (opts, i, info)          # the path under test is the one that ran
                if i == 0 and opts["spill_lanes"] >= 5:
                    assert info["lanes_last_launch"] > 1000, (opts, i, info)     # ... and had something to do on the deep windows
    # off by option, and off for shallow tiles by default
    with MandelbrotDevice(0) as dev:
        dev.set_option("spill_first", 0)
        dev.compute_view(views[0][0], views[0][1], want_bytes=False, precision=precision)
        assert dev.spill_info()["launches"] == 0
    with MandelbrotDevice(0) as dev:
        dev.compute_view(views[0][0], 4096, want_bytes=False, precision=precision)      # 2^14 blocks: under MBK_OPT_SPILL_MIN_BLOCKS
        assert dev.spill_info()["launches"] == 0
        dev.set_option("spill_min_blocks", 0)
        dev.compute_view(views[0][0], 1500, want_bytes=False, precision=precision)      # mrd below MBK_OPT_SPILL_MIN_MRD
        assert dev.spill_info()["launches"] == 0
        dev.compute_view(views[0][0], 2048, want_bytes=False, precision=precision)
        assert dev.spill_info()["launches"] == 1


def test_units_order_every_output_set_and_shape(oracle):
    """Order 3 ("units", csrc/mbk_units.h): the light blocks of eight block columns are one workgroup that runs them through
    the light path.  Every output instantiation (counts / bytes / both) x fp64 / fp32 x cycle test on / off through
    mbk_view_launch on device buffers, on views that exercise what the unit path must get right: rows that are no multiple
    of 8 and a window that stops short of the view (ragged edges -> class M), a window with an offset into a larger view,
    units in which the probe misses a long-lived block (the antenna: y = 0 midway between two probe rows), a tile with no
    set in it, a tile that is all set, and mrd at the light path's step limits; under both uneven deals across the XCDs."""
    import torch
    from distributedmandelbrot_amd import MandelbrotDevice
    cases = [
        (View(-2.0, -1.5, 3.0, 3.0, 2048, 2048), None, 300),
        (View(-2.0, -1.5, 3.0, 3.0, 2048, 1203), None, 200),                      # ragged last block row
        (View(-2.0, -2.0, 4.0, 4.0, 4096, 4096), (512, 1024, 2048, 1024), 150),   # a window inside a larger view
        (View(-2.0, -0.0125, 1.75, 0.025, 4096, 1027), None, 400),                # the antenna between probe rows
        (View(-2.0, -2.0, 1.0, 1.0, 2048, 1024), None, 1000),                     # all exterior: V units only
        (View(-0.2, -0.1, 0.2, 0.2, 1024, 1024), None, 120),                      # all interior: H only
        (View(-0.755, 0.10, 0.02, 0.02, 1024, 1024), None, 70),                   # boundary-rich: mostly M
        (View(-2.0, -1.5, 3.0, 3.0, 1536, 1024), None, 66),                       # mrd just above the 2 x probe depth gate
    ]
    with MandelbrotDevice(0) as dev:
        dev.set_option("order", 3)
        dev.set_option("units_min_light", 0)      # every window through the units kernel, also the ones it is not the default for
        for cyc in (1, 0):
            dev.set_option("cycle_detect", cyc)
            # the deal across the XCDs: shares that follow the time stamps of the launches before (they move while this
            # loop runs), then a fixed uneven deal
            dev.set_option("xcd_balance", 1 if cyc else 2)
            for view, window, mrd in cases:
                col0, row0, ncols, nrows = window if window else (0, 0, view.width, view.height)
                for precision in ("f64", "f32"):
                    oc, ob, total = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, view.width, view.height, mrd,
                                                window=window, precision=precision)
                    # the host API with bytes only (what a DataChunk asks for): the units kernel adds up the statistics itself
                    # and writes no int32 counts
                    _, hb, st = dev.compute_view(view, mrd, window=window, want_counts=False, kernel="group", precision=precision)
                    assert np.array_equal(hb, ob), (view, window, mrd, precision, cyc)
                    assert st.pixel_iterations == total and st.never_pixels == int((oc == 0).sum()), (view, window, mrd, precision, cyc)
                    assert st.all_bytes_zero == bool((ob == 0).all()) and st.all_bytes_one == bool((ob == 1).all())
                    for want_c, want_b in ((True, False), (False, True), (True, True)):
                        dc = torch.full((nrows * ncols,), -9, dtype=torch.int32, device="cuda:0") if want_c else None
                        db = torch.full((nrows * ncols,), 77, dtype=torch.uint8, device="cuda:0") if want_b else None
                        torch.cuda.synchronize()
                        dev.launch_view(view, mrd, window=window, d_counts=dc.data_ptr() if want_c else 0,
                                        d_bytes=db.data_ptr() if want_b else 0, stream=torch.cuda.current_stream().cuda_stream,
                                        kernel="group", precision=precision)
                        torch.cuda.synchronize()
                        tag = (view, window, mrd, precision, cyc, want_c, want_b)
                        if want_c:
                            got = dc.cpu().numpy().reshape(nrows, ncols)
                            assert np.array_equal(got, oc), (tag, int((got != oc).sum()))
                        if want_b:
                            assert np.array_equal(db.cpu().numpy().reshape(nrows, ncols), ob), tag


def test_xcd_shares_follow_solitary_strict_launches():
    """MBK_OPT_XCD_BALANCE = 1 (csrc/mbk_units.h, mbk_api.hip: xcd_shares_update): launches without the cycle test that have the
    chip to themselves leave 72 time stamps in pinned memory, and the host reads them at the next launch -- the feedback loop is
    alive (stamps arrive whole and carry the launch number), the shares stay a distribution inside their clamp, and the
    counts are what they were under the even deal; launches WITH the cycle test neither leave stamps nor move the shares."""
    import torch
    from distributedmandelbrot_amd import MandelbrotDevice
    view, mrd = View(-2.0, -1.5, 3.0, 3.0, 4096, 4096), 1000
    d = torch.zeros(4096 * 4096, dtype=torch.int32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    with MandelbrotDevice(0) as dev:
        dev.set_option("cycle_detect", 0)
        dev.set_option("xcd_balance", 0)
        dev.launch_view(view, mrd, d_counts=d.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        even = d.clone()
        base = dev.xcd_shares()
        assert base["last_launch_read"] == 0 and base["shares"] == [0.125] * 8
        dev.set_option("xcd_balance", 1)
        for _ in range(12):
            d.fill_(-5)
            dev.launch_view(view, mrd, d_counts=d.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            assert torch.equal(d, even)
        s = dev.xcd_shares()
        assert s["units_launches"] == base["units_launches"] + 12
        assert s["last_launch_read"] >= s["units_launches"] - 2, s      # every launch but the newest has been read
        assert abs(sum(s["shares"]) - 1.0) < 1e-3 and all(0.105 < f < 0.145 for f in s["shares"]), s
        dev.set_option("cycle_detect", 1)
        for _ in range(4):
            dev.launch_view(view, mrd, d_counts=d.data_ptr(), stream=stream)
            torch.cuda.synchronize()
        assert torch.equal(d, even)
        s2 = dev.xcd_shares()
        assert s2["shares"] == s["shares"] and s2["last_launch_read"] == s["last_launch_read"], (s, s2)


def test_quantiser_every_count_on_device(gpu, oracle):
    """The device divides by multiplying with a host reciprocal (mbk_kernels.h: quantise); check it
    against the reference's float form (WorkerCUDA.py:96-98, restated by the oracle) for EVERY count of
    many mrd, including the 64-bit path (mrd >= 2^23)."""
    from oracle.oracle import numpy_quantise
    for mrd in [1, 2, 3, 7, 255, 256, 257, 1000, 1024, 4999, 5000, 10000, 50000, 65535, 65536, 999983,
                2 ** 23 - 1, 2 ** 23, 2 ** 23 + 1, 2 ** 24 + 3, 2 ** 31 - 1]:
        if mrd <= 2 ** 22:
            counts = np.arange(mrd, dtype=np.int32)
        else:   # every count near both ends and around each byte boundary, plus a dense random sample
            k = np.arange(1, 257, dtype=np.int64) * mrd // 256
            edge = np.concatenate([k + d for d in range(-3, 4)])
            rs = np.random.RandomState(mrd % 1000)
            counts = np.unique(np.clip(np.concatenate([np.arange(4096), mrd - 1 - np.arange(4096), edge,
                                                       rs.randint(0, mrd, 1 << 20)]), 0, mrd - 1)).astype(np.int32)
        got = gpu.quantise_counts(counts, mrd)
        assert np.array_equal(got, numpy_quantise(counts, mrd)), mrd


def test_launches_on_many_streams_have_private_scratch(gpu, oracle):
    """Helper-kernel scratch is kept per stream (round 1 shared a ring of 8 across streams): 12 launches in
    flight on 12 torch streams, scan / group / refill mixed, every result bit-exact."""
    import torch
    view, mrd = View(-0.755, 0.10, 0.02, 0.02, 1024, 768), 1500
    oc, _, _ = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, 1024, 768, mrd, want_bytes=False)
    streams = [torch.cuda.Stream() for _ in range(12)]
    outs = [torch.full((768 * 1024,), -7, dtype=torch.int32, device="cuda:0") for _ in streams]
    torch.cuda.synchronize()
    for rep in range(3):
        for i, (s, o) in enumerate(zip(streams, outs)):
            gpu.launch_view(view, mrd, d_counts=o.data_ptr(), stream=s.cuda_stream,
                            kernel=["scan", "group", "refill"][(i + rep) % 3])
    torch.cuda.synchronize()
    for o in outs:
        assert np.array_equal(o.cpu().numpy().reshape(768, 1024), oc)


def test_render_view_into_pinned_image_two_slots(gpu, oracle):
    """render_view DMAs each band straight into its rows of a caller-supplied pinned image, two bands in
    flight on the device (mbk_view_submit / mbk_wait); the same image through plain numpy memory too."""
    from distributedmandelbrot_amd.sharding import render_view
    view, mrd = View(-0.755, 0.10, 0.02, 0.02, 1024, 1000), 900
    oc, ob, total = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, 1024, 1000, mrd)
    pc, pb = gpu.pinned_empty((1000, 1024), np.int32), gpu.pinned_empty((1000, 1024), np.uint8)
    pc[...] = -1
    c, b, per = render_view([gpu], view, mrd, band_rows=72, out_counts=pc, out_bytes=pb)
    assert c is pc and b is pb and np.array_equal(pc, oc) and np.array_equal(pb, ob)
    assert per[0]["bands"] == 14 and per[0]["pixel_iterations"] == total
    c2, b2, _ = render_view([gpu], view, mrd, band_rows=128, kernel="group")
    assert np.array_equal(c2, oc) and np.array_equal(b2, ob)


def test_pipelined_worker_on_gpu_against_fake_distributer(gpu, golden):
    """worker.run_pipelined on the real device: lease / compute / send overlapped, two tiles in flight;
    the server ends up with the reference-made golden bytes for every tile."""
    from distributedmandelbrot_amd import worker
    from fake_distributer import FakeDistributer
    with FakeDistributer([(4, 256)]) as srv:
        n = worker.run_pipelined("127.0.0.1", srv.port, device=gpu, log=lambda *a: None, senders=2, max_tiles=7)
        assert n == 7 and srv.wait_completed(7, timeout=60)
        for key, w in (("4_256_0_0", (4, 256, 0, 0)), ("4_256_1_2", (4, 256, 1, 2))):
            assert hashlib.sha256(srv.completed[w].tobytes()).hexdigest() == str(golden[f"full/{key}/bytes_sha256"])


def test_default_kernel_choice_any_arrival_order(oracle):
    """The default kernel is chosen per launch from a 256-pixel host probe of the window (round 3; rounds 1-2 used
    what the previous launch on the stream had reported): light tiles -> "scan", heavy tiles -> "group".  Whatever
    it picks, and in whatever order tiles arrive, results are bit-exact; a threshold of 0 / 65536 forces either
    path."""
    from distributedmandelbrot_amd import MandelbrotDevice
    heavy = (View(-0.2, -0.1, 0.2, 0.2, 1024, 1024), 300)      # inside the cardioid: every block heavy
    light = (View(-2.0, -2.0, 1.0, 1.0, 1024, 1024), 300)      # far exterior: nothing deferred
    mixed = (View(-2.0, -1.5, 3.0, 3.0, 1024, 1024), 300)
    want = {k: oracle.view(v.start_r, v.start_i, v.range_r, v.range_i, v.width, v.height, m, want_bytes=False)[0]
            for k, (v, m) in (("heavy", heavy), ("light", light), ("mixed", mixed))}
    cases = {"heavy": heavy, "light": light, "mixed": mixed}
    for share in (655, 0, 65536):
        with MandelbrotDevice(0) as dev:
            dev.set_option("heavy_share", share)
            for name in ["light", "heavy", "heavy", "light", "mixed", "mixed", "heavy", "light", "light"]:
                v, m = cases[name]
                c, _, _ = dev.compute_view(v, m, want_bytes=False)
                assert np.array_equal(c, want[name]), (share, name)


def test_lazy_uniform_skips_the_copy_only_for_uniform_tiles(gpu, golden):
    """MBK_LAZY_UNIFORM: the stats say whether the tile is uniform, and only then may the host buffer be left alone.  An
    all-exterior tile (host probe: every pixel gone within 4 steps) is not copied; an all-interior one is reported "Never"
    (since round 5 its copy is enqueued at submit, so the buffer is either untouched or all zero); a boundary tile and an
    all-exterior tile with non-uniform bytes are copied as usual; a tile wholly outside |c| = 2 costs no GPU work at all."""
    buf = gpu.pinned_empty((16777216,), np.uint8)
    buf[:] = 7
    gpu.submit_datachunk(0, 20, 1024, 7, 9, buf, lazy_uniform=True)           # golden: all 16 777 216 pixels in the set
    st = gpu.wait(0)
    assert st.all_bytes_zero and not st.all_bytes_one and ((buf == 7).all() or (buf == 0).all())
    gpu.submit_datachunk(1, 4, 256, 0, 0, buf, lazy_uniform=True)             # all exterior, but bytes 1..3: not uniform
    st = gpu.wait(1)
    assert not st.all_bytes_zero and not st.all_bytes_one
    assert hashlib.sha256(buf.tobytes()).hexdigest() == str(golden["full/4_256_0_0/bytes_sha256"])
    buf[:] = 7
    gpu.submit_datachunk(2, 16, 1024, 0, 0, buf, lazy_uniform=True)           # far corner, outside |c| = 2: answered on the host
    st = gpu.wait(2)
    assert st.all_bytes_one and (buf == 7).all() and st.never_pixels == 0 and st.pixel_iterations == 16777216
    assert st.kernel_ms == 0.0 and st.d2h_ms == 0.0 and st.rle_runs == 1
    gpu.submit_datachunk(3, 16, 1024, 2, 4, buf, lazy_uniform=True)           # reaches inside the circle, counts 1 and 2: byte 1
    st = gpu.wait(3)
    assert st.all_bytes_one and (buf == 7).all() and st.never_pixels == 0 and st.kernel_ms > 0.0 and st.d2h_ms < 0.05
    assert st.pixel_iterations == 33545871


def test_lazy_uniform_agrees_with_the_plain_path_on_a_whole_level(gpu, oracle):
    """Every tile of pyramid level 16 (mrd 1024 and 256) and the ring of level 40 through MBK_LAZY_UNIFORM with all slots in
    flight against the synchronous path: identical statistics, identical bytes wherever the tile is not uniform, and the tiles
    answered on the host (no kernel) are exactly those wholly outside |c| = 2 -- whose bytes the GPU agrees are all 1.  mrd 255
    (byte 2 outside the circle) and mrd 2 take the GPU path."""
    import ctypes as C
    from distributedmandelbrot_amd import _lib as L
    nslots = gpu.SLOTS
    pins = [gpu.pinned_empty((16777216,), np.uint8) for _ in range(nslots)]
    ref = gpu.pinned_empty((16777216,), np.uint8)
    lib = L.load()

    def outside(level, ir, ii):
        sr, si, rng = oracle.geometry(level, ir, ii)
        cv = L.mbk_view(sr, si, rng, rng, 4096, 4096, 0, 0, 4096, 4096)
        out = C.c_int(-1)
        assert lib.mbk_view_outside_circle(C.byref(cv), 0, C.byref(out)) == L.MBK_OK
        return bool(out.value)

    ring40 = [(ir, ii) for ir in range(40) for ii in range(40)
              if 3.2 < (min(abs(-2 + 0.1 * ir), abs(-2 + 0.1 * (ir + 1))) ** 2 + min(abs(-2 + 0.1 * ii), abs(-2 + 0.1 * (ii + 1))) ** 2) < 4.6]
    jobs = [(16, 1024, [(ir, ii) for ir in range(16) for ii in range(16)]), (16, 256, [(ir, ii) for ir in (0, 1, 5, 15) for ii in range(16)]),
            (40, 1024, ring40[::3]), (16, 255, [(0, 0), (15, 15), (3, 0)]), (16, 2, [(0, 0), (7, 7)])]
    for level, mrd, tiles in jobs:
        n, host_answered, got = len(tiles), 0, {}
        for i in range(n + nslots):
            if i >= nslots:
                j = i - nslots
                st = gpu.wait(j % nslots)
                got[tiles[j]] = (st, None if (st.all_bytes_zero or st.all_bytes_one) else pins[j % nslots].copy())
            if i < n:
                gpu.submit_datachunk(i % nslots, level, mrd, *tiles[i], pins[i % nslots], lazy_uniform=True)
        for t in tiles:
            st, data = got[t]
            _, _, want = gpu.datachunk(level, mrd, *t, out_bytes=ref)
            assert (st.pixel_iterations, st.never_pixels, st.all_bytes_zero, st.all_bytes_one, st.rle_runs) == \
                   (want.pixel_iterations, want.never_pixels, want.all_bytes_zero, want.all_bytes_one, want.rle_runs), (level, mrd, t)
            if data is not None:
                assert np.array_equal(data, ref), (level, mrd, t)
            skipped = st.kernel_ms == 0.0 and st.d2h_ms == 0.0
            assert skipped == (outside(level, *t) and mrd >= 256), (level, mrd, t)
            if skipped:
                host_answered += 1
                assert want.all_bytes_one and (ref == 1).all() and want.pixel_iterations == 16777216
        if (level, mrd) == (16, 1024):
            assert host_answered == 32
        if level == 40:
            assert 0 < host_answered < n


CYCLE_VIEWS = [
    (View(-0.2, -0.1, 0.2, 0.2, 256, 256), 5000),        # inside the main cardioid: every orbit settles on a fixed point
    (View(-1.1, -0.1, 0.2, 0.2, 200, 200), 3000),        # period-2 bulb around c = -1 (c = -1 itself: 0, -1, 0, -1, ...)
    (View(-0.16, 0.70, 0.08, 0.08, 160, 160), 4000),     # period-3 bulb
    (View(0.20, -0.05, 0.10, 0.10, 128, 128), 2000),     # the cusp at 1/4: parabolic, hardly any orbit becomes periodic
    (View(-2.0, -1.5, 3.0, 3.0, 512, 512), 3000),        # the whole set
    (View(-1.79, -0.03, 0.06, 0.06, 96, 96), 6000),      # the period-3 minibrot on the real axis
    (View(0.0, 0.0, 0.0, 0.0, 9, 9), 100000),            # c = 0 everywhere: fixed from the first step (step-0 linspace)
    (View(-0.5, -0.5, 0.5, 0.5, 517, 203), 1000),        # ragged edges + pinned end points around interior blocks
]


_ORACLE_MEMO = {}


def _oracle_view_memo(oracle, view, mrd, precision):
    key = (view, mrd, precision)
    if key not in _ORACLE_MEMO:
        _ORACLE_MEMO[key] = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, view.width, view.height,
                                        mrd, precision=precision)
    return _ORACLE_MEMO[key]


@pytest.mark.parametrize("kernel", ["default", "group", "scan"])
@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_cycle_detection_is_bit_exact(oracle, kernel, precision):
    """MBK_OPT_CYCLE_DETECT retires a pixel as 'never escapes' when its (zr, zi) bit pattern repeats.  Same counts as
    the strict loop (WorkerCUDA.py:39-68 runs such a pixel to mrd-1 and returns 0) on views made of set interior,
    for both settings, for every group size with a cycle test, and for mrd values around the loops' trip limits."""
    from distributedmandelbrot_amd import MandelbrotDevice
    for cyc, opts in ((1, {}), (1, {"group_steps": 8}), (0, {}), (1, {"cycle_window": 0}), (1, {"cycle_window": 65536, "group_steps": 8})):
        with MandelbrotDevice(0) as dev:
            dev.set_option("cycle_detect", cyc)
            for k, v in opts.items():
                dev.set_option(k, v)
            cases = list(CYCLE_VIEWS) if not opts else CYCLE_VIEWS[:3]
            if cyc:
                cases += [(View(-0.5, -0.5, 0.5, 0.5, 100, 100), m) for m in (9, 10, 24, 25, 26, 40, 41, 42, 56, 57, 58, 72, 73, 105)]
            for view, mrd in cases:
                c, b, st = dev.compute_view(view, mrd, kernel=kernel, precision=precision)
                oc, ob, total = _oracle_view_memo(oracle, view, mrd, precision)
                assert np.array_equal(c, oc), (view, mrd, kernel, precision, cyc, opts, int((c != oc).sum()))
                assert np.array_equal(b, ob) and st.pixel_iterations == total and st.never_pixels == int((oc == 0).sum())


def test_cycle_detection_smooth(gpu, oracle):
    """The smooth entry point with the cycle test on (the default): the value of a retired pixel is 0 like any
    never-escaping pixel's."""
    view, mrd = View(-0.3, -0.2, 0.5, 0.4, 200, 160), 2500
    nu, c, _ = gpu.compute_view_smooth(view, mrd, kernel="group")
    onu, oc = oracle.view_smooth(view.start_r, view.start_i, view.range_r, view.range_i, view.width, view.height, mrd)
    assert np.array_equal(c, oc)
    assert np.array_equal(nu == 0.0, oc == 0) and np.allclose(nu, onu, rtol=0, atol=1e-12 * mrd)
    assert gpu.get_option("cycle_detect") == 1


def test_stats_reduction_vector_and_scalar_paths(gpu, oracle):
    """The per-tile statistics (pixel-iterations, never-escaped, all-0 / all-1, RLE runs: what DataChunk.cs:82,87 and
    DataChunkSerializer.cs:56-100 would find by scanning) come from one reduction kernel with a four-pixels-per-lane
    form for aligned buffers and a scalar form otherwise: both against numpy, for sizes around the group width,
    for offsets that break the alignment, and for byte patterns with runs ending on every position of a word."""
    import torch
    from oracle.serializer import rle_runs
    rs = np.random.RandomState(9)
    big = torch.from_numpy(rs.randint(0, 50, size=(1 << 20) + 64).astype(np.int32)).to("cuda:0")
    host = big.cpu().numpy()
    for off in (0, 1, 2, 3, 4, 5):
        for n in (1, 3, 4, 5, 1023, 1024, 1025, 1027, 4099, (1 << 20) + 3):
            for mrd in (2, 1000):
                st = gpu.reduce_counts(big.data_ptr() + 4 * off, n, mrd)
                ref = host[off:off + n].astype(np.int64)
                assert st.pixel_iterations == int(np.where(ref > 0, ref, mrd - 1).sum()) and st.never_pixels == int((ref == 0).sum()), (off, n, mrd)
    # byte statistics go through the view entry points: ragged widths put run boundaries on every byte of a word
    for w, h, mrd in ((517, 203, 40), (1024, 257, 300), (1031, 129, 9), (64, 16, 100), (4099, 8, 50)):
        view = View(-2.0, -1.25, 2.5, 2.5, w, h)
        c, b, st = gpu.compute_view(view, mrd)
        oc, ob, total = oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, w, h, mrd)
        assert np.array_equal(b, ob) and st.pixel_iterations == total and st.never_pixels == int((oc == 0).sum())
        assert st.rle_runs == len(rle_runs(ob.reshape(-1))[0]), (w, h, mrd)
        assert st.all_bytes_zero == bool((ob == 0).all()) and st.all_bytes_one == bool((ob == 1).all())


# ------------------------------------------------------------------------------------------------------
# Round 3
# ------------------------------------------------------------------------------------------------------

def test_default_kernel_choice_is_history_independent():
    """The kernel of a default launch depends on its window only: an all-exterior tile costs the same (the light
    pass, tens of microseconds) whether the launch before it was light or heavy.  Rounds 1-2 picked from the previous
    launch's report and sent the first light tile after a heavy one down the one-workgroup-per-block path (~70 us)."""
    from distributedmandelbrot_amd import MandelbrotDevice
    with MandelbrotDevice(0) as dev:
        buf = dev.pinned_empty((16777216,), np.uint8)
        for _ in range(3):
            dev.datachunk(4, 1000, 0, 0, out_bytes=buf)                       # warm
        after_light = min(dev.datachunk(4, 1000, 0, 0, out_bytes=buf)[2].kernel_ms for _ in range(5))
        after_heavy = []
        for _ in range(5):
            dev.datachunk(4, 1000, 1, 2, out_bytes=buf)                       # half of it inside the set
            after_heavy.append(dev.datachunk(4, 1000, 0, 0, out_bytes=buf)[2].kernel_ms)
        assert after_light < 0.060, after_light
        assert min(after_heavy) < 1.3 * after_light + 0.005, (after_light, after_heavy)


def test_slot0_users_refuse_while_a_tile_is_in_flight(gpu, golden):
    """mbk_view_compute_smooth, mbk_serialize_last, mbk_quantise_counts and the synchronous compute calls work on
    slot 0's buffers, events and reduction scratch: with a tile in flight there they return MBK_ERR_INVALID and
    leave it alone; mbk_reduce_counts on a caller stream has scratch of its own and may run.  The tile then
    arrives intact."""
    import torch
    from distributedmandelbrot_amd import MbkError
    buf = gpu.pinned_empty((16777216,), np.uint8)
    small = View(-2.0, -1.5, 3.0, 3.0, 64, 64)
    gpu.datachunk(4, 256, 0, 0, out_bytes=buf)          # so that serialize_last has a "last tile"
    d = torch.from_numpy(np.arange(4096, dtype=np.int32)).to("cuda:0")
    gpu.submit_datachunk(0, 10, 1024, 3, 5, buf)        # boundary-rich golden tile: a few hundred microseconds
    with pytest.raises(MbkError):
        gpu.compute_view_smooth(small, 100)
    with pytest.raises(MbkError):
        gpu.serialize_last()
    with pytest.raises(MbkError):
        gpu.quantise_counts(np.arange(10, dtype=np.int32), 100)
    with pytest.raises(MbkError):
        gpu.compute_view(small, 100)
    st_r = gpu.reduce_counts(d.data_ptr(), 4096, 5000, stream=torch.cuda.current_stream().cuda_stream)
    assert st_r.pixel_iterations == 4999 + sum(range(1, 4096)) and st_r.never_pixels == 1
    st = gpu.wait(0)
    assert hashlib.sha256(buf.tobytes()).hexdigest() == str(golden["full/10_1024_3_5/bytes_sha256"])
    assert st.never_pixels == int(golden["full/10_1024_3_5/zeros"])
    gpu.compute_view_smooth(small, 100)                 # and slot 0 is usable again
    byts, _, _ = gpu.datachunk(4, 256, 0, 0)
    from oracle.serializer import serialize
    assert gpu.serialize_last()[0] == serialize(byts)


CFG3_LEVEL, CFG3_MRD, CFG3_IR, CFG3_II = 800000, 10000, 251270, 426364   # cfg3 as DataChunks


@pytest.fixture(scope="module")
def cfg3_chunks_oracle(oracle):
    """The four DataChunk tiles of level 800 000 around BASELINE cfg3's centre (range 5e-6 each), from the 8-lane
    AVX-512 evaluation of the oracle on the tiles' own geometry (WorkerCUDA.py:75-78)."""
    if not oracle.have_avx512():
        pytest.skip("host without AVX-512: the scalar oracle needs minutes for these tiles")
    out = {}
    for dr in (0, 1):
        for di in (0, 1):
            sr, si, rng = oracle.geometry(CFG3_LEVEL, CFG3_IR + dr, CFG3_II + di)
            out[(dr, di)] = oracle.view_avx512(sr, si, rng, rng, 4096, 4096, CFG3_MRD)[0]
    return out


def test_cfg3_as_datachunk_tiles(gpu, cfg3_chunks_oracle):
    """BASELINE cfg3 in its DataChunk form: 2 x 2 tiles of level 800 000, mrd 10000, through mbk_datachunk -- counts
    and bytes against the oracle, and the reference's tile quirk on DEVICE output: np.linspace includes the end
    point, so tile (ir, ii)'s last column is computed at start_r + range and tile (ir+1, ii)'s first at
    -2 + range*(ir+1).  Wherever those two roundings agree (always at power-of-two levels; at level
    800 000 for some index pairs only -- the reference's geometry, WorkerCUDA.py:75-78, not ours) the shared edge
    must be identical on the device; where they differ by an ulp the edges are different samples."""
    got = {}
    for (dr, di), oc in cfg3_chunks_oracle.items():
        byts, counts, st = gpu.datachunk(CFG3_LEVEL, CFG3_MRD, CFG3_IR + dr, CFG3_II + di, want_counts=True)
        counts = counts.reshape(4096, 4096)
        assert np.array_equal(counts, oc), ((dr, di), int((counts != oc).sum()))
        ob = ((oc.astype(np.int64) * 256 + CFG3_MRD - 1) // CFG3_MRD).astype(np.uint8)
        assert np.array_equal(byts.reshape(4096, 4096), ob)
        assert st.pixel_iterations == int(np.where(oc > 0, oc, CFG3_MRD - 1).astype(np.int64).sum())
        got[(dr, di)] = counts
    from oracle.oracle import numpy_geometry
    shared = 0
    for k in (0, 1):
        r0, i0 = numpy_geometry(CFG3_LEVEL, CFG3_IR, CFG3_II + k), numpy_geometry(CFG3_LEVEL, CFG3_IR + k, CFG3_II)
        r1, i1 = numpy_geometry(CFG3_LEVEL, CFG3_IR + 1, CFG3_II + k), numpy_geometry(CFG3_LEVEL, CFG3_IR + k, CFG3_II + 1)
        if r0[0] + r0[2] == r1[0]:      # real neighbours: last column == first column
            assert np.array_equal(got[(0, k)][:, -1], got[(1, k)][:, 0])
            shared += 1
        if i0[1] + i0[2] == i1[1]:      # imaginary neighbours: last row == first row
            assert np.array_equal(got[(k, 0)][-1, :], got[(k, 1)][0, :])
            shared += 1
    assert shared >= 2                  # (the two real-axis pairs coincide at these indices)


def test_power_of_two_level_tiles_share_their_edges(gpu):
    """At a power-of-two level every tile boundary is exact, so adjacent DataChunk tiles share their edge column /
    row sample for sample: level 16 around the seahorse valley, on device output."""
    a = gpu.datachunk(16, 1024, 5, 8, want_counts=True)[1].reshape(4096, 4096)
    b = gpu.datachunk(16, 1024, 6, 8, want_counts=True)[1].reshape(4096, 4096)
    c = gpu.datachunk(16, 1024, 5, 9, want_counts=True)[1].reshape(4096, 4096)
    assert np.array_equal(a[:, -1], b[:, 0]) and np.array_equal(a[-1, :], c[0, :])
    assert 0 < int((a == 0).sum()) < a.size          # a tile that holds part of the set, not a trivial one


def test_cfg3_datachunks_through_a_farm_of_two_feeders(gpu, cfg3_chunks_oracle):
    """The same four tiles leased by the stand-in Distributer to TWO pipelined feeders (run_farm; both on GPU 0 --
    any number of clients may pull from the one hand-out loop, Distributer.cs:335-392): every tile arrives once,
    with the oracle's bytes."""
    from distributedmandelbrot_amd import worker
    from distributedmandelbrot_amd.server import Distributer

    class Window(Distributer):          # hand out exactly the 2 x 2 tiles around cfg3's centre
        def _next_needed(self):
            import time as _t
            now = _t.monotonic()
            self.leases = [(w, t) for w, t in self.leases if now < t]
            leased = {w for w, _ in self.leases} | set(self.receiving)
            for dr in (0, 1):
                for di in (0, 1):
                    w = (CFG3_LEVEL, CFG3_MRD, CFG3_IR + dr, CFG3_II + di)
                    if (w[0], w[2], w[3]) not in self.completed and w not in leased:
                        return w
            return None

    got = {}

    class Keep:
        def completed(self):
            return []

        def save_chunk(self, level, ir, ii, payload):
            got[(ir - CFG3_IR, ii - CFG3_II)] = np.array(payload, copy=True)

    with Window([(CFG3_LEVEL, CFG3_MRD)], store=Keep()) as dist:
        done = worker.run_farm("127.0.0.1", dist.port, devices=[0, 0], log=lambda *a: None)
        assert sum(done) == 4 and dist.received == 4 and not dist.rejected
    for key, oc in cfg3_chunks_oracle.items():
        ob = ((oc.astype(np.int64) * 256 + CFG3_MRD - 1) // CFG3_MRD).astype(np.uint8)
        assert np.array_equal(got[key].reshape(4096, 4096), ob), key


def test_prepass_overlap_transitions_and_regrowth(oracle):
    """The dispatch-order pre-pass runs on an auxiliary stream into two alternating lists (MBK_OPT_PREPASS_OVERLAP):
    back-to-back launches of changing size (the lists are re-allocated when a window has more blocks), switching
    the option between launches, two caller streams at once -- every result bit-exact."""
    import torch
    from distributedmandelbrot_amd import MandelbrotDevice
    views = [(View(-2.0, -1.5, 3.0, 3.0, 1024, 1024), 300), (View(-0.755, 0.10, 0.02, 0.02, 1536, 1100), 700),
             (View(-2.0, -1.5, 3.0, 3.0, 2048, 1024), 200), (View(-0.2, -0.1, 0.2, 0.2, 1024, 1032), 150)]
    want = [oracle.view(v.start_r, v.start_i, v.range_r, v.range_i, v.width, v.height, m, want_bytes=False)[0] for v, m in views]
    with MandelbrotDevice(0) as dev:
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        plan = [(rep * 5 + rep // 3) % len(views) for rep in range(12)]
        outs = [(i, torch.full((views[i][0].width * views[i][0].height,), -3, dtype=torch.int32, device="cuda:0")) for i in plan]
        torch.cuda.synchronize()      # the fills ran on torch's current stream; the launches below go to other streams
        for rep, (i, o) in enumerate(outs):
            v, m = views[i]
            dev.set_option("prepass_overlap", 0 if rep in (4, 5, 9) else 1)
            dev.launch_view(v, m, d_counts=o.data_ptr(), stream=streams[rep % 2].cuda_stream, kernel="group")
        torch.cuda.synchronize()
        for i, o in outs:
            v, _ = views[i]
            assert np.array_equal(o.cpu().numpy().reshape(v.height, v.width), want[i]), i


def test_pci_bus_id_and_queue_tiles_of_the_bench(gpu, oracle):
    """What bench.py's multi-GPU line reports per rank, and the tiles of its --shard queue job: windows of one
    grid x 4096 view are bit-identical to the oracle's evaluation of the same window."""
    import re
    assert re.fullmatch(r"[0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9a-fA-F]", gpu.pci_bus_id())
    grid = 8
    qview = View(-2.0, -1.5, 3.0, 3.0, 4096 * grid, 4096 * grid)
    for tr, ti, rows in ((0, 0, 64), (3, 4, 48), (5, 3, 32), (7, 7, 64)):
        window = (tr * 4096, ti * 4096 + 1000, 4096, rows)
        c, _, st = gpu.compute_view(qview, 1000, window=window, want_bytes=False)
        oc, _, total = oracle.view(qview.start_r, qview.start_i, qview.range_r, qview.range_i, qview.width, qview.height,
                                   1000, window=window, want_bytes=False)
        assert np.array_equal(c, oc) and st.pixel_iterations == total, (tr, ti)


def test_native_worker_loop_on_gpu(gpu, golden):
    """mbk_worker_run (the worker loop inside libmbk_hip.so: WorkerCUDA.py:111-184 pipelined) on the real device
    against the restated Distributer: the server ends up with the reference-made golden bytes, uniform tiles
    included (level 4 has four all-exterior corner tiles that are NOT uniform and none that is; level 16 mrd 1024
    tile (0,0) is all "Immediate")."""
    from distributedmandelbrot_amd import worker
    from distributedmandelbrot_amd.server import Distributer
    from fake_distributer import FakeDistributer
    with FakeDistributer([(4, 256)]) as srv:
        n = worker.run_native("127.0.0.1", srv.port, device=gpu, log=lambda *a: None, senders=3, max_tiles=7)
        assert n == 7 and srv.wait_completed(7, timeout=60)
        for key, w in (("4_256_0_0", (4, 256, 0, 0)), ("4_256_1_2", (4, 256, 1, 2))):
            assert hashlib.sha256(srv.completed[w].tobytes()).hexdigest() == str(golden[f"full/{key}/bytes_sha256"])
    got = {}

    class Keep:
        def completed(self):
            return []

        def save_chunk(self, level, ir, ii, payload):
            got[(ir, ii)] = (int(payload[0]), bool((payload == payload[0]).all()), hashlib.sha256(payload.tobytes()).hexdigest())

    with Distributer([(16, 1024)], store=Keep()) as dist:
        n = worker.run_native("127.0.0.1", dist.port, device=gpu, log=lambda *a: None, senders=4, max_tiles=40)
        import time
        end = time.time() + 60
        while dist.received < 40 and time.time() < end:
            time.sleep(0.01)
        assert n == 40 and dist.received == 40
    assert got[(0, 0)][:2] == (1, True)                 # far corner: every pixel escapes at step 1 -> byte 1
    # a tile the Python loop computes the same way (same library call underneath): spot-check one mixed tile
    ir, ii = next(k for k, v in got.items() if not v[1])
    ref, _, _ = gpu.datachunk(16, 1024, ir, ii)
    assert hashlib.sha256(ref.tobytes()).hexdigest() == got[(ir, ii)][2]
    with pytest.raises(Exception):
        worker.run_native("127.0.0.1", 1, device=gpu, log=lambda *a: None)   # nobody listens on port 1: MBK_ERR_NET


# Windows whose 16 x 16 host probe sees nothing that outlives the light pass although they hold part of the set: the
# antenna along the real axis (y = 0 lies midway between two probe rows, 0.1 away, where every pixel escapes at step 3)
# with its minibrots; the same in a window with ragged edges and pinned end points; a far-exterior ragged window.
NEEDLE_VIEWS = [
    (View(-2.1, -1.6, 0.3, 3.2, 1024, 2049), 500),
    (View(-2.1, -1.6, 0.3, 3.2, 515, 1031), 300),
    (View(-2.0, -2.0, 1.0, 1.0, 1001, 1003), 200),
]


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_scan_finishes_in_place_what_the_probe_missed(oracle, precision):
    """MBK_OPT_SCAN_INLINE (round 3): when the host probe of a window finds no pixel that outlives the light pass, pass 1
    of "scan" finishes its unfinished blocks itself and pass 2 is not launched.  The probe is a heuristic: these windows
    hold hundreds of in-set blocks it does not see (and edge blocks, which never take the light path).  Bit-exact for
    every output set, with and without the cycle test, and identical to the two-pass form."""
    from distributedmandelbrot_amd import MandelbrotDevice
    for inline, cyc, strip in ((1, 1, 1), (1, 0, 1), (1, 1, 0), (0, 1, 1)):   # (strip: MBK_OPT_SCAN_STRIP, round 5 -- every window
        with MandelbrotDevice(0) as dev:                                        #  here is wide enough for row strips)
            dev.set_option("scan_inline", inline)
            dev.set_option("cycle_detect", cyc)
            dev.set_option("scan_strip", strip)
            for view, mrd in NEEDLE_VIEWS:
                oc, ob, total = _oracle_view_memo(oracle, view, mrd, precision)
                assert view.width == 1001 or int((oc == 0).sum()) > 300      # the antenna windows do hold part of the set
                for kernel in ("scan", "default"):
                    c, b, st = dev.compute_view(view, mrd, kernel=kernel, precision=precision)
                    assert np.array_equal(c, oc), (view, mrd, kernel, precision, inline, cyc, strip, int((c != oc).sum()))
                    assert np.array_equal(b, ob) and st.pixel_iterations == total and st.never_pixels == int((oc == 0).sum())
                # bytes only (what a DataChunk asks for: with the fused statistics no int32 count is written at all) / counts only
                _, b2, st2 = dev.compute_view(view, mrd, kernel="scan", precision=precision, want_counts=False)
                c3, _, st3 = dev.compute_view(view, mrd, kernel="scan", precision=precision, want_bytes=False)
                assert np.array_equal(b2, ob) and np.array_equal(c3, oc)
                for st in (st2, st3):
                    assert st.pixel_iterations == total and st.never_pixels == int((oc == 0).sum()), (view, precision, inline, cyc, strip)
                runs = 1 + int((ob.ravel()[1:] != ob.ravel()[:-1]).sum())
                assert st2.rle_runs == runs and st2.all_bytes_zero == bool((ob == 0).all()) and st2.all_bytes_one == bool((ob == 1).all())


# All-exterior windows (the host probe finds nothing that outlives the light pass -> the finish-in-place form of "scan") in
# the shapes the row strips must get right: whole DataChunk-sized tiles, widths that are no multiple of 64 (a ragged last
# strip), a window at an offset inside a larger view, a window that holds the pinned last sample of both axes, the smallest
# mrd the light path serves (4 steps) and one whose quantiser maps counts 1..4 to different bytes.
STRIP_VIEWS = [
    (View(-2.0, -2.0, 1.0, 1.0, 4096, 4096), 1024, None),                      # DataChunk (4,0,0) at the reference's mrd
    (View(-2.0, -2.0, 4.0, 0.9, 2048, 700), 300, None),                        # counts 1 .. 8+, one pixel of the set (the needle's tip)
    (View(-2.0, -2.0, 4.0, 0.9, 4120, 300), 6, None),                          # 64 strips + 24 columns; 5 steps: 658 pixels end as 0
    (View(-2.0, -2.0, 4.0, 0.9, 520, 77), 5, None),                            # the fewest steps the light path serves
    (View(-2.0, -2.0, 4.0, 4.0, 4096, 4096), 500, (0, 0, 4096, 900)),
    (View(-2.0, -2.0, 4.0, 4.0, 8192, 8192), 700, (1003, 205, 4997, 1301)),    # odd offsets, odd sizes
    (View(-2.0, -2.0, 4.0, 4.0, 4100, 4100), 64, (700, 3200, 3400, 900)),      # holds the pinned last sample of both axes
]


@pytest.mark.parametrize("precision", ["f64", "f32"])
def test_scan_row_strips_every_output_set_and_shape(oracle, precision):
    """MBK_OPT_SCAN_STRIP (round 5): in the finish-in-place form of "scan" a wave's region is 64 x 1 pixels instead of an 8x8
    block (same asm loop, same arithmetic: which lane holds which pixel), so that a store instruction writes one contiguous
    piece of a row.  Bit-exact against the oracle for every output set, with the option on and off, with and without the
    cycle test; the statistics a bytes-only launch sums in the kernel included."""
    from distributedmandelbrot_amd import MandelbrotDevice
    expected = [oracle.view(view.start_r, view.start_i, view.range_r, view.range_i, view.width, view.height, mrd, window=window,
                            precision=precision) for view, mrd, window in STRIP_VIEWS]
    for strip, cyc in ((1, 1), (0, 1), (1, 0)):
        with MandelbrotDevice(0) as dev:
            dev.set_option("scan_strip", strip)
            dev.set_option("cycle_detect", cyc)
            for (view, mrd, window), (oc, ob, total) in zip(STRIP_VIEWS, expected):
                never = int((oc == 0).sum())
                for kernel in ("scan", "default"):
                    c, b, st = dev.compute_view(view, mrd, window=window, kernel=kernel, precision=precision)
                    assert np.array_equal(c, oc), (view, mrd, window, kernel, precision, strip, cyc, int((c != oc).sum()))
                    assert np.array_equal(b, ob) and st.pixel_iterations == total and st.never_pixels == never
                _, b2, st2 = dev.compute_view(view, mrd, window=window, kernel="scan", precision=precision, want_counts=False)
                c3, _, st3 = dev.compute_view(view, mrd, window=window, kernel="scan", precision=precision, want_bytes=False)
                assert np.array_equal(b2, ob) and np.array_equal(c3, oc), (view, mrd, window, precision, strip, cyc)
                for st in (st2, st3):
                    assert st.pixel_iterations == total and st.never_pixels == never, (view, mrd, window, precision, strip, cyc)
                runs = 1 + int((ob.ravel()[1:] != ob.ravel()[:-1]).sum())
                assert st2.rle_runs == runs and st2.all_bytes_zero == bool((ob == 0).all()) and st2.all_bytes_one == bool((ob == 1).all())

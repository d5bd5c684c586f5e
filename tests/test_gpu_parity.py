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

KERNELS = ["default", "simple", "asm", "refill", "group"]


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


@pytest.mark.parametrize("kernel", ["default", "asm", "group"])
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


@pytest.mark.parametrize("kernel", ["default", "asm", "group"])
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

"""CPU tests of the oracle itself: before the oracle judges the GPU it is pinned against
(1) vectors produced by EXECUTING the reference's own Python source (tests/golden/make_golden.py),
(2) an independent numpy restatement, (3) numpy.linspace itself, (4) hand-checked points."""
import hashlib

import numpy as np
import pytest

from oracle.oracle import (numpy_axis, numpy_escape, numpy_geometry, numpy_quantise, numpy_view,
                           pixel_iterations)


def test_golden_small_windows_match_c_oracle(oracle, golden):
    for name in golden["small/names"]:
        sr, si, rng, n, mrd = golden[f"small/{name}/params"]
        n, mrd = int(n), int(mrd)
        counts, _, total = oracle.view(sr, si, rng, rng, n, n, mrd, want_bytes=mrd > 0)
        ref = golden[f"small/{name}/counts"]
        assert np.array_equal(counts, ref), name
        assert total == pixel_iterations(ref, mrd), name
        # the reference's own coordinate arrays (np.linspace inside gen_arrays)
        assert np.array_equal(oracle.axis(sr, rng, n), golden[f"small/{name}/axis_r"]), name
        assert np.array_equal(oracle.axis(si, rng, n), golden[f"small/{name}/axis_i"]), name


def test_golden_points_match_c_oracle(oracle, golden):
    for (cr, ci, mrd), ref in zip(golden["points/inputs"], golden["points/counts"]):
        assert oracle.escape(cr, ci, int(mrd)) == int(ref), (cr, ci, mrd)


def test_hand_checked_points(oracle):
    # c = 0 never escapes; c = -2: z1 = 4-2 = 2, |z|^2 = 4 -> escapes at 1 under '>=' (WorkerCUDA.py:65);
    # c = 2+2i: |z1|^2 huge -> 1; mrd <= 1 -> range(1, mrd) is empty -> 0.
    assert oracle.escape(0.0, 0.0, 1000) == 0
    assert oracle.escape(-2.0, 0.0, 1000) == 1
    assert oracle.escape(2.0, 2.0, 1000) == 1
    assert oracle.escape(-2.0, 0.0, 1) == 0
    assert oracle.escape(-2.0, 0.0, 0) == 0
    assert oracle.escape(-2.0, 0.0, 2) == 1
    assert oracle.escape(0.25, 0.0, 100000) == 0      # parabolic point: never reaches 4
    assert oracle.escape(0.26, 0.0, 1000) == 29


def test_golden_full_tiles_match_c_oracle(oracle, golden):
    """Full 4096x4096 tiles produced by the reference's unmodified process_workload."""
    for key in golden["full/names"]:
        level, mrd, ir, ii = (int(x) for x in golden[f"full/{key}/params"])
        counts, byts, total = oracle.datachunk(level, mrd, ir, ii)
        assert hashlib.sha256(byts.tobytes()).hexdigest() == str(golden[f"full/{key}/bytes_sha256"]), key
        assert hashlib.sha256(counts.astype("<i4").tobytes()).hexdigest() == \
            str(golden[f"full/{key}/counts_sha256"]), key
        assert np.array_equal(byts[::64, ::64], golden[f"full/{key}/bytes_sub64"])
        assert np.array_equal(counts[::64, ::64], golden[f"full/{key}/counts_sub64"])
        assert int((counts == 0).sum()) == int(golden[f"full/{key}/zeros"])


@pytest.mark.parametrize("level,ir,ii", [(1, 0, 0), (3, 1, 2), (4, 1, 2), (7, 6, 0), (10, 3, 5),
                                          (20, 7, 9), (20, 19, 19), (800000, 251270, 426364),
                                          (4294967295, 4294967294, 1)])
def test_geometry_and_axis_match_python_and_numpy(oracle, level, ir, ii):
    sr, si, rng = oracle.geometry(level, ir, ii)
    assert (sr, si, rng) == numpy_geometry(level, ir, ii)
    for start in (sr, si):
        assert np.array_equal(oracle.axis(start, rng, 4096), np.linspace(start, start + rng, 4096))
    # the last sample is the stop value start+range itself (endpoint=True), so adjacent tiles share
    # their edge column/row up to the rounding of start+range
    assert oracle.axis(sr, rng, 4096)[-1] == sr + rng


@pytest.mark.parametrize("n", [1, 2, 3, 7, 64, 513, 4096, 8192])
def test_axis_matches_linspace_generic(oracle, n):
    rs = np.random.RandomState(n)
    for _ in range(20):
        start = float(rs.uniform(-2, 2))
        rng = float(10.0 ** rs.uniform(-12, 0.6))
        assert np.array_equal(oracle.axis(start, rng, n), numpy_axis(start, rng, n))
    # numpy's step == 0 fallback (range so small that step underflows to zero / delta == 0)
    assert np.array_equal(oracle.axis(1.0, 0.0, n), numpy_axis(1.0, 0.0, n))
    assert np.array_equal(oracle.axis(0.0, 5e-324, n), numpy_axis(0.0, 5e-324, n))


def test_c_oracle_matches_numpy_oracle_on_seeded_views(oracle):
    rs = np.random.RandomState(1234)
    cases = [(-2.0, -1.5, 3.0, 3.0, 96, 80, 256), (-0.743648, 0.131820, 1e-5, 1e-5, 40, 40, 3000)]
    for _ in range(6):
        cr, ci = rs.uniform(-1.6, 0.4), rs.uniform(-1.1, 1.1)
        span = 10.0 ** rs.uniform(-6, 0)
        cases.append((cr, ci, span, span * rs.uniform(0.5, 2), int(rs.randint(1, 90)),
                      int(rs.randint(1, 90)), int(rs.randint(1, 700))))
    for sr, si, rr, ri, w, h, mrd in cases:
        c, b, total = oracle.view(sr, si, rr, ri, w, h, mrd)
        c2, b2 = numpy_view(sr, si, rr, ri, w, h, mrd)
        assert np.array_equal(c, c2) and np.array_equal(b, b2)
        assert total == pixel_iterations(c, mrd)
        # windows are bit-identical to the corresponding slice of the whole view
        if w > 3 and h > 3:
            win = (1, 2, w - 2, h - 3)
            cw, bw, _ = oracle.view(sr, si, rr, ri, w, h, mrd, window=win)
            assert np.array_equal(cw, c[2:2 + h - 3, 1:1 + w - 2]) and np.array_equal(bw, b[2:2 + h - 3, 1:1 + w - 2])


@pytest.mark.parametrize("mrd", [1, 2, 3, 255, 256, 257, 1000, 1024, 4095, 10000, 65535, 1000003])
def test_quantiser_integer_form(oracle, mrd):
    """WorkerCUDA.py:96-98 (float ceil + uint8 wrap) == ((count*256 + mrd-1) // mrd) & 0xFF, the form
    the GPU kernel uses, for every legal count."""
    counts = np.arange(0, mrd, dtype=np.int64)
    if mrd > 70000:
        rs = np.random.RandomState(mrd)
        counts = np.unique(np.concatenate([counts[:3000], counts[-3000:], rs.randint(0, mrd, 50000)]))
    ref = numpy_quantise(counts.astype(np.int32), mrd)
    integer = (((counts * 256 + mrd - 1) // mrd) & 0xFF).astype(np.uint8)
    assert np.array_equal(ref, integer)
    sample = counts[:: max(1, len(counts) // 500)]
    assert [oracle.quantise(int(c), mrd) for c in sample] == [int(x) for x in integer[:: max(1, len(counts) // 500)]]


def test_quantiser_wrap_examples(oracle):
    # for mrd = 1000 counts 997..999 wrap to byte 0
    assert [oracle.quantise(c, 1000) for c in (0, 1, 3, 4, 996, 997, 999)] == [0, 1, 1, 2, 255, 0, 0]
    # huge mrd (64-bit path on the GPU)
    for mrd in (2 ** 23, 2 ** 24 + 1, 2 ** 31 - 1):
        for c in (0, 1, mrd // 3, mrd - 2, mrd - 1):
            assert oracle.quantise(c, mrd) == ((c * 256 + mrd - 1) // mrd) & 0xFF


def test_numpy_escape_matches_scalar(oracle):
    rs = np.random.RandomState(7)
    cr = rs.uniform(-2, 0.6, 300)
    ci = rs.uniform(-1.2, 1.2, 300)
    v = numpy_escape(cr, ci, 400)
    assert [oracle.escape(a, b, 400) for a, b in zip(cr, ci)] == v.tolist()


def test_f32_variant_c_oracle_matches_numpy_float32(oracle):
    """BASELINE cfg4's fp32 variant is not in the reference; its definition is the strict-binary32
    restatement.  The C and numpy versions must agree bit-for-bit before either judges the GPU."""
    rs = np.random.RandomState(44)
    cases = [(-2.0, -1.5, 3.0, 3.0, 120, 90, 300), (-0.755, 0.10, 0.02, 0.02, 64, 64, 2000)]
    for _ in range(4):
        cr, ci = rs.uniform(-1.6, 0.4), rs.uniform(-1.1, 1.1)
        span = 10.0 ** rs.uniform(-4, 0)
        cases.append((cr, ci, span, span, int(rs.randint(1, 80)), int(rs.randint(1, 80)), int(rs.randint(2, 900))))
    for sr, si, rr, ri, w, h, mrd in cases:
        c, b, total = oracle.view(sr, si, rr, ri, w, h, mrd, precision="f32")
        xr, xi = numpy_axis(sr, rr, w), numpy_axis(si, ri, h)
        c2 = numpy_escape(xr[None, :], xi[:, None], mrd, dtype=np.float32).reshape(h, w)
        assert np.array_equal(c, c2)
        assert np.array_equal(b, numpy_quantise(c2, mrd)) and total == pixel_iterations(c, mrd)
    assert oracle.lib.mbo_escape_f32(-2.0, 0.0, 100) == 1 and oracle.lib.mbo_escape_f32(0.0, 0.0, 100) == 0


def test_smooth_oracle_consistency(oracle):
    """cfg5 oracle: integer part equals the parity count; value matches a numpy evaluation of the formula."""
    sm, c = oracle.view_smooth(-2.0, -1.5, 3.0, 3.0, 96, 64, 500)
    c0, _, _ = oracle.view(-2.0, -1.5, 3.0, 3.0, 96, 64, 500, want_bytes=False)
    assert np.array_equal(c, c0)
    assert (sm[c == 0] == 0).all() and (sm[c > 0] > c[c > 0]).all()
    # c = 2+2i escapes at n=1 with z1 = (2+2i)^2 + (2+2i) = 2 + 10i -> |z|^2 = 104
    v = oracle.lib.mbo_escape_smooth
    import ctypes as C
    v.restype = C.c_double; v.argtypes = [C.c_double, C.c_double, C.c_int32, C.c_void_p]
    assert abs(v(2.0, 2.0, 10, None) - (2.0 - np.log2(0.5 * np.log(104.0)))) < 1e-15


def test_avx512_baseline_is_bit_identical_to_scalar_oracle(oracle):
    if not oracle.have_avx512():
        pytest.skip("host CPU has no AVX-512")
    for sr, si, rr, ri, w, h, mrd in [(-2.0, -1.5, 3.0, 3.0, 203, 77, 300), (-0.755, 0.10, 0.02, 0.02, 64, 64, 2000),
                                      (-2.0, -2.0, 4.0, 4.0, 9, 5, 2), (0.3, 0.3, 1e-3, 1e-3, 17, 3, 1)]:
        c, _, total = oracle.view(sr, si, rr, ri, w, h, mrd, want_bytes=False)
        c2, total2 = oracle.view_avx512(sr, si, rr, ri, w, h, mrd)
        assert np.array_equal(c, c2) and total == total2


def test_quantiser_reciprocal_form():
    """The device's division-free quantiser (csrc/mbk_kernels.h: q = trunc(fma(x, fl(1/mrd), 2^-30)) with
    x = count*256 + mrd - 1) restated with exact rationals: float(Fraction) is the correctly rounded fma.
    Every count for small mrd; byte boundaries, both ends and a random sample for large mrd < 2^23."""
    from fractions import Fraction
    bias = Fraction(1, 2 ** 30)

    def device_form(count: int, mrd: int) -> int:
        rcp = Fraction(1.0 / mrd)                       # the host's correctly rounded reciprocal, exactly
        x = count * 256 + mrd - 1
        return int(float(x * rcp + bias)) & 0xFF        # one rounding, like v_fma_f64; trunc; uint8 wrap

    rs = np.random.RandomState(11)
    for mrd in [1, 2, 3, 7, 100, 255, 256, 257, 1000, 1023, 1024, 5000]:
        for count in range(mrd):
            assert device_form(count, mrd) == ((count * 256 + mrd - 1) // mrd) & 0xFF, (count, mrd)
    for mrd in [10000, 50000, 65535, 999983, 2 ** 22 + 1, 2 ** 23 - 1]:
        ks = np.arange(1, 257, dtype=np.int64) * mrd // 256
        counts = set(int(c) for d in range(-2, 3) for c in ks + d) | set(range(64)) | set(range(mrd - 64, mrd)) \
            | set(int(c) for c in rs.randint(0, mrd, 3000))
        for count in counts:
            if 0 <= count < mrd:
                assert device_form(count, mrd) == ((count * 256 + mrd - 1) // mrd) & 0xFF, (count, mrd)


def test_contraction_whatif_is_a_different_function(oracle):
    """oracle.view_contracted (default CUDA FMA contraction applied to WorkerCUDA.py:50-62) is a what-if used
    for DESIGN.md's sensitivity table, NOT the parity target: it agrees with the strict oracle on almost every
    pixel but not on all of them (scripts/contraction_table.py: up to 3 870 of 16.8 M pixels per golden tile)."""
    strict, _, _ = oracle.view(-0.75, 0.09, 0.02, 0.02, 512, 512, 2000, want_bytes=False)
    fused = oracle.view_contracted(-0.75, 0.09, 0.02, 0.02, 512, 512, 2000)
    diff = int((strict != fused).sum())
    assert 0 < diff < 0.05 * strict.size


def test_cycle_retirement_claim(oracle):
    """What the GPU kernels' cycle test (MBK_OPT_CYCLE_DETECT) relies on, checked on the CPU model of it
    (oracle.view_cycle): stopping a pixel as 'never escapes' when its (zr, zi) bit pattern repeats gives exactly
    the counts of the strict loop (WorkerCUDA.py:39-68) -- on interior-heavy views, for several check periods --
    while running fewer steps than mrd-1 for pixels of the set."""
    cases = [(-2.0, -1.5, 3.0, 3.0, 400, 400, 1000), (-0.2, -0.1, 0.2, 0.2, 96, 96, 5000),
             (-1.1, -0.1, 0.2, 0.2, 96, 96, 3000), (-0.16, 0.70, 0.08, 0.08, 96, 96, 4000),
             (0.20, -0.05, 0.10, 0.10, 64, 64, 2000), (-0.743648, 0.131820, 1e-5, 1e-5, 64, 64, 3000)]
    saved = 0
    for sr, si, rr, ri, w, h, mrd in cases:
        strict, _, _ = oracle.view(sr, si, rr, ri, w, h, mrd, want_bytes=False)
        # (8, 8, 32) = the kernels' schedule (MBK_OPT_CYCLE_WINDOW = 32, round 5; 0 = the doubled windows of rounds 2-4)
        for first, check, wcap in ((8, 8, 32), (8, 8, 0), (8, 16, 0), (0, 1, 0), (8, 32, 5), (3, 7, 65536), (0, 8, 1)):
            counts, executed = oracle.view_cycle(sr, si, rr, ri, w, h, mrd, first=first, check=check, window_cap=wcap)
            assert np.array_equal(counts, strict), (sr, si, mrd, first, check, wcap)
            ref_steps = np.where(strict > 0, strict, mrd - 1)
            assert (executed <= ref_steps).all() and (executed[strict > 0] == strict[strict > 0]).all()
            saved += int((ref_steps - executed).sum())
    assert saved > 0
    # c = 0 and c = -1: exactly periodic from the start
    for cr in (0.0, -1.0):
        counts, executed = oracle.view_cycle(cr, 0.0, 0.0, 0.0, 1, 1, 100000)
        assert counts[0, 0] == 0 and executed[0, 0] < 100


def test_outside_circle_escapes_at_step_one(oracle):
    """The claim behind MBK_LAZY_UNIFORM's short cut (csrc/mbk_api.hip: submit_view, view_outside_circle2): every sample c
    with |c|^2 >= 4 (1 + m), m = 1e-8 (binary64) / 1e-4 (binary32), gets count 1 from calc_mb_value (WorkerCUDA.py:54-63)
    for any mrd >= 2, so a window that lies wholly outside that circle is an "Immediate" chunk (every byte
    ceil(256 / mrd) = 1 for mrd >= 256) and needs no GPU.

    In exact arithmetic: z1 = c^2 + c = c (c + 1), |z1| = |c| |c + 1| >= |c| (|c| - 1); with |c| = 2 (1 + d), d >= 0:
    |z1| >= 2 (1 + d) (1 + 2 d) >= 2 (1 + 3 d), |z1|^2 >= 4 (1 + 6 d).  |c|^2 >= 4 (1 + m) means d >= m / 2 - m^2 / 8, so
    |z1|^2 >= 4 (1 + 2.9 m).  The reference computes |z1|^2 with ten individually rounded operations on operands of like
    sign or with a result bounded away from cancellation only in the products -- the one subtraction zr^2 - zi^2 can cancel,
    but then its ABSOLUTE error is at most 2 u max(zr^2, zi^2) <= 2 u |c|^2, against |z1| >= |c| (|c| - 1) >= |c|^2 / 2: a
    relative error of each component of z1 below 8 u in all, of |z1|^2 below 40 u (u = 2^-53 resp. 2^-24): 4.4e-15 resp.
    2.4e-6, four to seven orders of magnitude under 2.9 m.  Checked here with exact rationals on the adversarial arc
    around c = -2 (where |c + 1| is smallest), on the whole circle, and on random points; then the library's own host
    predicate (mbk_view_outside_circle: no device needed) is checked against the oracle on random views and on every
    DataChunk tile of levels 1..24."""
    from fractions import Fraction
    from distributedmandelbrot_amd import _lib as L
    import ctypes as C

    rs = np.random.RandomState(11)

    def exact_norm2_z1(cr, ci):
        cr, ci = Fraction(cr), Fraction(ci)
        zr, zi = cr * cr - ci * ci + cr, 2 * cr * ci + ci
        return zr * zr + zi * zi

    for m, esc, cast in ((1e-8, oracle.escape, float), (1e-4, oracle.escape_f32, np.float32)):
        r = 2.0 * np.sqrt(1.0 + m) * (1.0 + 1e-12)
        angles = np.concatenate([np.pi + np.linspace(-1e-3, 1e-3, 201), np.pi + rs.uniform(-0.2, 0.2, 300),
                                 np.linspace(0.0, 2.0 * np.pi, 721), rs.uniform(0.0, 2.0 * np.pi, 2000)])
        radii = np.concatenate([np.full(angles.size // 2, r), r * (1.0 + 10.0 ** rs.uniform(-9, 2, angles.size - angles.size // 2))])
        worst = None
        for th, rad in zip(angles, radii):
            cr, ci = float(cast(rad * np.cos(th))), float(cast(rad * np.sin(th)))
            if cr * cr + ci * ci < 4.0 * (1.0 + m):
                continue      # rounding of the polar form put the sample inside the margin: not a sample the predicate admits
            n2 = exact_norm2_z1(cr, ci)
            assert n2 >= Fraction(4) * (1 + Fraction(2.8 * m)), (cr, ci, float(n2))
            worst = n2 if worst is None or n2 < worst else worst
            for mrd in (2, 3, 1000):
                assert esc(cr, ci, mrd) == 1, (cr, ci, mrd)
        assert float(worst) < 4.0 * (1.0 + 4.0 * m)     # the arc around c = -2 was really sampled

    lib = L.load()

    def outside(view, window=None, f32=False):
        col0, row0, ncols, nrows = window if window else (0, 0, view[4], view[5])
        cv = L.mbk_view(view[0], view[1], view[2], view[3], view[4], view[5], col0, row0, ncols, nrows)
        out = C.c_int(-1)
        assert lib.mbk_view_outside_circle(C.byref(cv), L.MBK_PRECISION_F32 if f32 else 0, C.byref(out)) == L.MBK_OK
        return bool(out.value)

    # every DataChunk tile of levels 1..24: the predicate is the geometry it claims to be (recomputed here from np.linspace's
    # axis), it never admits a tile with a count other than 1 (a tile comes closest to the circle on its border), and it
    # finds what the geometry promises: the corners of [-2, 2]^2 outside the inscribed circle -- 32 of level 16's 256 tiles
    # (NOT the 192 "Immediate" tiles of that level at mrd 1024: the other 160 hold counts 1..4, which all quantise to byte 1,
    # and their statistics need the exact counts)
    admitted = {}
    for level in range(1, 25):
        k = 0
        for ir in range(level):
            for ii in range(level):
                sr, si, rng = oracle.geometry(level, ir, ii)
                xs, ys = numpy_axis(sr, rng, 4096), numpy_axis(si, rng, 4096)
                dx = 0.0 if xs[0] <= 0 <= xs[-1] else min(abs(xs[0]), abs(xs[-1]))
                dy = 0.0 if ys[0] <= 0 <= ys[-1] else min(abs(ys[0]), abs(ys[-1]))
                want = dx * dx + dy * dy >= 4.0 * (1.0 + 1e-8)
                assert outside((sr, si, rng, rng, 4096, 4096)) == want, (level, ir, ii)
                if not want:
                    continue
                k += 1
                for win in ((0, 0, 4096, 1), (0, 4095, 4096, 1), (0, 0, 1, 4096), (4095, 0, 1, 4096)):
                    c, _, _ = oracle.view(sr, si, rng, rng, 4096, 4096, 1024, window=win, want_bytes=False)
                    assert (c == 1).all(), (level, ir, ii, win)
        admitted[level] = k
    assert admitted[16] == 32 and admitted[4] == 0 and admitted[8] == 4 and admitted[24] == 92, admitted
    # random views and windows, both precisions: admitted -> the oracle's counts are all 1
    hits = 0
    for _ in range(400):
        w, h = int(rs.randint(1, 300)), int(rs.randint(1, 300))
        span = 10.0 ** rs.uniform(-3, 0.7)
        ang, rad = rs.uniform(0, 2 * np.pi), 2.0 + 10.0 ** rs.uniform(-6, 0.5) * rs.choice([1.0, 1.0, -0.05])
        view = (rad * np.cos(ang) - rs.uniform(0, span), rad * np.sin(ang) - rs.uniform(0, span), span, span * rs.uniform(0.2, 2.0), w, h)
        col0, row0 = int(rs.randint(0, w)), int(rs.randint(0, h))
        window = (col0, row0, int(rs.randint(1, w - col0 + 1)), int(rs.randint(1, h - row0 + 1)))
        for f32 in (False, True):
            if outside(view, window, f32):
                hits += 1
                c, _, _ = oracle.view(*view[:4], w, h, 500, window=window, want_bytes=False, precision="f32" if f32 else "f64")
                assert (c == 1).all(), (view, window, f32)
    assert hits > 40, hits

/*
 * mandel_oracle.c -- CPU ORACLE for the Mandelbrot tile escape-time path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (distributedmandelbrot_amd/)
 * never imports, links or executes anything in oracle/.
 *
 * It restates, in strict IEEE-754 binary64 with every operation individually rounded
 * (compile with -ffp-contract=off, never -ffast-math), the reference's only implementation
 * of the hot path.  All citations are relative to /root/reference/ and
 * "WorkerCUDA.py" = DistributedMandelbrotWorkerCUDA/DistributedMandelbrotWorkerCUDA.py:
 *
 *   mbo_geometry   <- process_workload, WorkerCUDA.py:75-78  (== DataChunk.cs:32-33,59-66)
 *   mbo_axis       <- gen_arrays, WorkerCUDA.py:24-32 (np.linspace with endpoint; numpy is an
 *                     un-vendored, un-pinned dependency -- its published algorithm,
 *                     numpy/_core/function_base.py `linspace`, is restated here and was
 *                     checked bit-for-bit against numpy 2.2.6, see tests/test_oracle.py)
 *   mbo_escape     <- calc_mb_value, WorkerCUDA.py:39-68
 *   mbo_quantise   <- process_workload, WorkerCUDA.py:96-98
 *   mbo_view       <- gen_arrays' tile/repeat layout (WorkerCUDA.py:34-35) + the ufunc launch (:92)
 *   mbo_datachunk  <- process_workload as a whole, WorkerCUDA.py:70-100
 *
 * Parity pinning: the reference has no tests and no golden vectors.
 * The restatement is pinned instead against the reference's OWN Python source executed under
 * CPython (numba replaced by a scalar shim): tests/golden/make_golden.py writes
 * tests/golden/reference_vectors.npz.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MBO_MIN_AXIS (-2.0) /* WorkerCUDA.py:7, DataChunk.cs:14 */
#define MBO_MAX_AXIS (2.0)  /* WorkerCUDA.py:8, DataChunk.cs:15 */

/* WorkerCUDA.py:75-78.  Python: chunk_range = (MAX_AXIS - MIN_AXIS) / level  (int/int true
 * division -> double); start = MIN_AXIS + (chunk_range * index). */
void mbo_geometry(uint32_t level, uint32_t index_real, uint32_t index_imag,
                  double *start_r, double *start_i, double *range)
{
    volatile double chunk_range = (MBO_MAX_AXIS - MBO_MIN_AXIS) / (double)level;
    volatile double pr = chunk_range * (double)index_real;
    volatile double pi = chunk_range * (double)index_imag;
    *range = chunk_range;
    *start_r = MBO_MIN_AXIS + pr;
    *start_i = MBO_MIN_AXIS + pi;
}

/* np.linspace(start, start + range, num=n) as called at WorkerCUDA.py:24-32 (endpoint=True).
 * numpy: div = n-1; delta = stop - start; y = arange(n) as double; step = delta/div;
 *        if step == 0: y = (y/div)*delta  else: y = y*step;  y += start;  y[-1] = stop (n>1). */
void mbo_axis(double start, double range, uint32_t n, double *out)
{
    if (n == 0) return;
    volatile double stop = start + range;
    if (n == 1) { out[0] = start; return; }
    volatile double delta = stop - start;
    double div = (double)(n - 1);
    volatile double step = delta / div;
    for (uint32_t k = 0; k + 1 < n; ++k) {
        volatile double y;
        if (step == 0.0) {
            volatile double q = (double)k / div;
            y = q * delta;
        } else {
            y = (double)k * step;
        }
        out[k] = y + start;
    }
    out[n - 1] = stop;
}

/* calc_mb_value, WorkerCUDA.py:39-68.  z starts at c (:45); for i in range(1, mrd) (:47):
 * z = (z0*z0 - z1*z1, 2*z0*z1) (:50-53); z += c (:56-59); sqr_mag = z0*z0 + z1*z1 (:62);
 * if sqr_mag >= 4 return i (:65-66); return 0 (:68).
 * Written exactly as the source groups it: (2*z0)*z1, left to right. */
int32_t mbo_escape(double cr, double ci, int32_t mrd)
{
    double zr = cr, zi = ci;
    for (int32_t n = 1; n < mrd; ++n) {
        double a = zr * zr;
        double b = zi * zi;
        double t = a - b;
        double w = 2.0 * zr;
        double u = w * zi;
        zr = t + cr;
        zi = u + ci;
        double m0 = zr * zr;
        double m1 = zi * zi;
        double m = m0 + m1;
        if (m >= 4.0) return n;
    }
    return 0;
}

/* BASELINE config 4's fp32 variant (NOT in the reference; parity unpinned by it -- this function IS its
 * definition): the same loop in strict binary32.  The coordinates are the fp64 np.linspace values
 * rounded once to float. */
int32_t mbo_escape_f32(float cr, float ci, int32_t mrd)
{
    float zr = cr, zi = ci;
    for (int32_t n = 1; n < mrd; ++n) {
        float a = zr * zr;
        float b = zi * zi;
        float t = a - b;
        float w = 2.0f * zr;
        float u = w * zi;
        zr = t + cr;
        zi = u + ci;
        float m0 = zr * zr;
        float m1 = zi * zi;
        float m = m0 + m1;
        if (m >= 4.0f) return n;
    }
    return 0;
}

/* What-if variant ("contraction sensitivity"): the same loop with the FMA contraction a
 * CUDA toolchain applies by default to `a*b + c` patterns (numba -> NVVM, fmad on): the three
 * expressions of WorkerCUDA.py:50-62 each lose one rounding --
 *     z0*z0 - z1*z1   -> fma(z0, z0, -(z1*z1))
 *     (2*z0)*z1 + c1  -> fma(2*z0, z1, c1)
 *     z0*z0 + z1*z1   -> fma(z0, z0, z1*z1)
 * This is NOT the parity target (the reference's source, read strictly, is); it exists only to count how
 * many pixels of a tile would differ between this worker and a worker whose compiler contracted. */
int32_t mbo_escape_contracted(double cr, double ci, int32_t mrd)
{
    double zr = cr, zi = ci;
    for (int32_t n = 1; n < mrd; ++n) {
        double b = zi * zi;
        double t = fma(zr, zr, -b);
        double w = 2.0 * zr;
        double u = fma(w, zi, ci);
        zr = t + cr;
        zi = u;
        double m1 = zi * zi;
        double m = fma(zr, zr, m1);
        if (m >= 4.0) return n;
    }
    return 0;
}

uint64_t mbo_view_contracted(double start_r, double start_i, double range_r, double range_i,
                             uint32_t width, uint32_t height, int32_t mrd, int32_t *counts, int nthreads)
{
    double *xr = (double *)malloc(sizeof(double) * (width ? width : 1));
    double *xi = (double *)malloc(sizeof(double) * (height ? height : 1));
    mbo_axis(start_r, range_r, width, xr);
    mbo_axis(start_i, range_i, height, xi);
    uint64_t total = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : total)
#else
    (void)nthreads;
#endif
    for (int64_t r = 0; r < (int64_t)height; ++r)
        for (uint32_t c = 0; c < width; ++c) {
            int32_t cnt = mbo_escape_contracted(xr[c], xi[r], mrd);
            counts[(size_t)r * width + c] = cnt;
            total += cnt > 0 ? (uint64_t)cnt : (uint64_t)(mrd > 1 ? mrd - 1 : 0);
        }
    free(xr);
    free(xi);
    return total;
}

/* What-if model of the GPU kernels' cycle test (MBK_OPT_CYCLE_DETECT; NOT in the reference, and not an oracle:
 * tests use it to check the CLAIM the kernels rely on, and scripts/cycle_model.py to count the steps saved).
 * The reference loop as in mbo_escape; after the first `first` steps, every `check` steps the state (zr, zi) is
 * compared BITWISE with a saved state, which is replaced after 1, 2, 4, 8, ... comparisons (Brent).  The step
 * map is a function of those bits, so an exact repeat means the orbit is periodic, every state of the period
 * has already passed the bailout test, and the reference's loop would run to mrd-1 and return 0: the model
 * stops there.  Returns the count; *executed = steps actually run (== the reference's unless it stopped early). */
int32_t mbo_escape_cycle_w(double cr, double ci, int32_t mrd, int32_t first, int32_t check, uint32_t window_cap, int32_t *executed);
int32_t mbo_escape_cycle(double cr, double ci, int32_t mrd, int32_t first, int32_t check, int32_t *executed)
{
    return mbo_escape_cycle_w(cr, ci, mrd, first, check, 0u, executed);
}

/* window_cap (round 5, MBK_OPT_CYCLE_WINDOW): the saved state's window grows by a quarter (+ 1) while it is shorter than
 * window_cap comparisons and doubles from there on; 0 = always doubles (Brent, rounds 2-4).  Any schedule is exact. */
int32_t mbo_escape_cycle_w(double cr, double ci, int32_t mrd, int32_t first, int32_t check, uint32_t window_cap, int32_t *executed)
{
    double zr = cr, zi = ci, sr = 0.0, si = 0.0;
    int have = 0;
    uint32_t tc = 0, win = 1;
    int32_t n;
    for (n = 1; n < mrd; ++n) {
        double a = zr * zr;
        double b = zi * zi;
        double t = a - b;
        double w = 2.0 * zr;
        double u = w * zi;
        zr = t + cr;
        zi = u + ci;
        double m = zr * zr + zi * zi;
        if (m >= 4.0) {
            *executed = n;
            return n;
        }
        if (n >= first && check > 0 && (n - first) % check == 0) {
            uint64_t br, bi, bsr, bsi;
            __builtin_memcpy(&br, &zr, 8);
            __builtin_memcpy(&bi, &zi, 8);
            __builtin_memcpy(&bsr, &sr, 8);
            __builtin_memcpy(&bsi, &si, 8);
            if (!have) {  /* the kernels take the first reference state when the grouped loop starts */
                sr = zr, si = zi, have = 1;
            } else {
                if (br == bsr && bi == bsi) {
                    *executed = n;
                    return 0;
                }
                if (++tc >= win) sr = zr, si = zi, tc = 0, win += win < window_cap ? win / 4u + 1u : win;
            }
        }
    }
    *executed = mrd > 1 ? mrd - 1 : 0;
    return 0;
}

/* counts and executed steps of a whole view under the model above (row-parallel) */
void mbo_view_cycle_w(double start_r, double start_i, double range_r, double range_i, uint32_t width, uint32_t height,
                      int32_t mrd, int32_t first, int32_t check, uint32_t window_cap, int32_t *counts, int32_t *executed, int nthreads);
void mbo_view_cycle(double start_r, double start_i, double range_r, double range_i, uint32_t width, uint32_t height,
                    int32_t mrd, int32_t first, int32_t check, int32_t *counts, int32_t *executed, int nthreads)
{
    mbo_view_cycle_w(start_r, start_i, range_r, range_i, width, height, mrd, first, check, 0u, counts, executed, nthreads);
}

void mbo_view_cycle_w(double start_r, double start_i, double range_r, double range_i, uint32_t width, uint32_t height,
                      int32_t mrd, int32_t first, int32_t check, uint32_t window_cap, int32_t *counts, int32_t *executed, int nthreads)
{
    double *xr = (double *)malloc(sizeof(double) * (width ? width : 1));
    double *xi = (double *)malloc(sizeof(double) * (height ? height : 1));
    mbo_axis(start_r, range_r, width, xr);
    mbo_axis(start_i, range_i, height, xi);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#else
    (void)nthreads;
#endif
    for (int64_t r = 0; r < (int64_t)height; ++r)
        for (uint32_t c = 0; c < width; ++c)
            counts[(size_t)r * width + c] = mbo_escape_cycle_w(xr[c], xi[r], mrd, first, check, window_cap, &executed[(size_t)r * width + c]);
    free(xr);
    free(xi);
}

/* BASELINE config 5 (NOT in the reference): continuous escape-time value at the reference's bailout.
 * Runs the reference loop, keeps |z|^2 of the escaping step, nu = n + 1 - log2(0.5 * ln |z_n|^2); 0 if the
 * pixel never escapes.  *count_out receives the integer escape index. */
double mbo_escape_smooth(double cr, double ci, int32_t mrd, int32_t *count_out)
{
    double zr = cr, zi = ci;
    for (int32_t n = 1; n < mrd; ++n) {
        double a = zr * zr;
        double b = zi * zi;
        double t = a - b;
        double w = 2.0 * zr;
        double u = w * zi;
        zr = t + cr;
        zi = u + ci;
        double m0 = zr * zr;
        double m1 = zi * zi;
        double m = m0 + m1;
        if (m >= 4.0) {
            if (count_out) *count_out = n;
            return (double)n + 1.0 - log2(0.5 * log(m));
        }
    }
    if (count_out) *count_out = 0;
    return 0.0;
}

void mbo_view_smooth(double start_r, double start_i, double range_r, double range_i,
                     uint32_t width, uint32_t height, int32_t mrd, double *smooth, int32_t *counts)
{
    double *xr = (double *)malloc(sizeof(double) * (width ? width : 1));
    double *xi = (double *)malloc(sizeof(double) * (height ? height : 1));
    mbo_axis(start_r, range_r, width, xr);
    mbo_axis(start_i, range_i, height, xi);
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int64_t r = 0; r < (int64_t)height; ++r)
        for (uint32_t c = 0; c < width; ++c) {
            int32_t cnt;
            smooth[(size_t)r * width + c] = mbo_escape_smooth(xr[c], xi[r], mrd, &cnt);
            if (counts) counts[(size_t)r * width + c] = cnt;
        }
    free(xr);
    free(xi);
}

/* WorkerCUDA.py:96-98: out = (out.astype(float64) * 256) / mrd; ceil(out).astype(uint8).
 * 256.0 wraps to 0 (numpy's C cast on x86-64).  mrd == 0 would be 0/0 = NaN -> 0. */
uint8_t mbo_quantise(int32_t count, uint32_t mrd)
{
    if (mrd == 0) return 0;
    volatile double x = (double)count * 256.0;
    volatile double q = x / (double)mrd;
    return (uint8_t)(int64_t)ceil(q);
}

/*
 * One rectangular window [row0,row0+nrows) x [col0,col0+ncols) of a width x height view whose
 * axes are linspace(start_r, start_r+range_r, width) / linspace(start_i, start_i+range_i, height).
 * Output element (row-row0)*ncols + (col-col0): real is the fast axis (np.tile, WorkerCUDA.py:34),
 * imaginary the slow one (np.repeat, :35).  Either output pointer may be NULL.
 * Returns the number of pixel-iterations executed (count if count>0 else max(mrd-1,0)).
 */
static uint64_t mbo_view_impl(double start_r, double start_i, double range_r, double range_i,
                              uint32_t width, uint32_t height,
                              uint32_t col0, uint32_t row0, uint32_t ncols, uint32_t nrows,
                              int32_t mrd, int32_t *counts, uint8_t *bytes, int nthreads, int f32)
{
    double *xr = (double *)malloc(sizeof(double) * (width ? width : 1));
    double *xi = (double *)malloc(sizeof(double) * (height ? height : 1));
    mbo_axis(start_r, range_r, width, xr);
    mbo_axis(start_i, range_i, height, xi);
    uint64_t total = 0;
    int64_t cap = mrd > 1 ? (int64_t)mrd - 1 : 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : total)
#else
    (void)nthreads;
#endif
    for (int64_t r = 0; r < (int64_t)nrows; ++r) {
        double ci = xi[row0 + r];
        uint64_t acc = 0;
        for (uint32_t c = 0; c < ncols; ++c) {
            int32_t cnt = f32 ? mbo_escape_f32((float)xr[col0 + c], (float)ci, mrd)
                              : mbo_escape(xr[col0 + c], ci, mrd);
            size_t o = (size_t)r * ncols + c;
            if (counts) counts[o] = cnt;
            if (bytes) bytes[o] = mbo_quantise(cnt, (uint32_t)mrd);
            acc += cnt > 0 ? (uint64_t)cnt : (uint64_t)cap;
        }
        total += acc;
    }
    free(xr);
    free(xi);
    return total;
}

uint64_t mbo_view(double start_r, double start_i, double range_r, double range_i,
                  uint32_t width, uint32_t height,
                  uint32_t col0, uint32_t row0, uint32_t ncols, uint32_t nrows,
                  int32_t mrd, int32_t *counts, uint8_t *bytes, int nthreads)
{
    return mbo_view_impl(start_r, start_i, range_r, range_i, width, height, col0, row0, ncols, nrows, mrd,
                         counts, bytes, nthreads, 0);
}

uint64_t mbo_view_f32(double start_r, double start_i, double range_r, double range_i,
                      uint32_t width, uint32_t height,
                      uint32_t col0, uint32_t row0, uint32_t ncols, uint32_t nrows,
                      int32_t mrd, int32_t *counts, uint8_t *bytes, int nthreads)
{
    return mbo_view_impl(start_r, start_i, range_r, range_i, width, height, col0, row0, ncols, nrows, mrd,
                         counts, bytes, nthreads, 1);
}

/* process_workload, WorkerCUDA.py:70-100: one 4096 x 4096 DataChunk tile. */
uint64_t mbo_datachunk(uint32_t level, uint32_t mrd, uint32_t index_real, uint32_t index_imag,
                       int32_t *counts, uint8_t *bytes, int nthreads)
{
    double sr, si, range;
    mbo_geometry(level, index_real, index_imag, &sr, &si, &range);
    return mbo_view(sr, si, range, range, 4096, 4096, 0, 0, 4096, 4096, (int32_t)mrd, counts,
                    bytes, nthreads);
}

/*
 * "Best-effort CPU" baseline (bench.py only): the same strict arithmetic, 8 pixels at a time in AVX-512
 * registers -- explicit mul/add/sub intrinsics, never fmadd, so the counts are bit-identical to
 * mbo_escape.  Processes one row segment; lanes that escaped keep iterating harmlessly (their count is
 * frozen), the loop ends when all 8 are done or after mrd-1 steps.  Returns pixel-iterations.
 * Compiled only where the compiler supports the target attribute; selected at run time.
 */
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define MBO_HAVE_AVX512 1
__attribute__((target("avx512f"))) static uint64_t mbo_row_avx512(const double *xr, double ci, uint32_t ncols,
                                                                   int32_t mrd, int32_t *counts)
{
    uint64_t iters = 0;
    const __m512d four = _mm512_set1_pd(4.0), two = _mm512_set1_pd(2.0), vci = _mm512_set1_pd(ci);
    for (uint32_t c0 = 0; c0 < ncols; c0 += 8) {
        const uint32_t nv = ncols - c0 < 8 ? ncols - c0 : 8;
        const __mmask8 valid = (__mmask8)((1u << nv) - 1u);
        const __m512d vcr = _mm512_maskz_loadu_pd(valid, xr + c0);
        __m512d zr = vcr, zi = vci;
        __m512i cnt = _mm512_setzero_si512();
        __mmask8 live = valid;
        for (int32_t n = 1; n < mrd && live; ++n) {
            const __m512d a = _mm512_mul_pd(zr, zr), b = _mm512_mul_pd(zi, zi);
            const __m512d t = _mm512_sub_pd(a, b);
            const __m512d u = _mm512_mul_pd(_mm512_mul_pd(two, zr), zi);
            zr = _mm512_add_pd(t, vcr);
            zi = _mm512_add_pd(u, vci);
            const __m512d m = _mm512_add_pd(_mm512_mul_pd(zr, zr), _mm512_mul_pd(zi, zi));
            const __mmask8 esc = _mm512_mask_cmp_pd_mask(live, m, four, _CMP_GE_OQ);
            cnt = _mm512_mask_mov_epi64(cnt, esc, _mm512_set1_epi64(n));
            live &= (__mmask8)~esc;
        }
        long long tmp[8];
        _mm512_storeu_si512((void *)tmp, cnt);
        for (uint32_t k = 0; k < nv; ++k) {
            if (counts) counts[c0 + k] = (int32_t)tmp[k];
            iters += tmp[k] > 0 ? (uint64_t)tmp[k] : (uint64_t)(mrd > 1 ? mrd - 1 : 0);
        }
    }
    return iters;
}
#endif

int mbo_have_avx512(void)
{
#ifdef MBO_HAVE_AVX512
    return __builtin_cpu_supports("avx512f") ? 1 : 0;
#else
    return 0;
#endif
}

/* Whole view with the AVX-512 row kernel (counts optional).  Returns 0 if AVX-512 is unavailable. */
uint64_t mbo_view_avx512(double start_r, double start_i, double range_r, double range_i, uint32_t width,
                         uint32_t height, int32_t mrd, int32_t *counts, int nthreads)
{
#ifdef MBO_HAVE_AVX512
    if (!mbo_have_avx512()) return 0;
    double *xr = (double *)malloc(sizeof(double) * (width ? width : 1));
    double *xi = (double *)malloc(sizeof(double) * (height ? height : 1));
    mbo_axis(start_r, range_r, width, xr);
    mbo_axis(start_i, range_i, height, xi);
    uint64_t total = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads) reduction(+ : total)
#endif
    for (int64_t r = 0; r < (int64_t)height; ++r)
        total += mbo_row_avx512(xr, xi[r], width, mrd, counts ? counts + (size_t)r * width : NULL);
    free(xr);
    free(xi);
    return total;
#else
    (void)start_r; (void)start_i; (void)range_r; (void)range_i; (void)width; (void)height; (void)mrd;
    (void)counts; (void)nthreads;
    return 0;
#endif
}

int mbo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

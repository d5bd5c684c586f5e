// mbk_kernels.h -- device code of libmbk_hip.so: escape-time kernels for gfx950 (MI355X / CDNA4).
//
// What is computed (reference = DistributedMandelbrotWorkerCUDA.py, "W.py"):
//   coordinates   x[k] = fl(fl(k*step) + start), x[n-1] = stop          np.linspace, W.py:24-32
//   escape loop   z = c; for n in 1..mrd-1: z = z*z + c; if |z|^2 >= 4 return n; return 0   W.py:39-68
//   quantiser     byte = ceil(count*256/mrd) mod 256                     W.py:96-98
//
// Bit-exactness rules for everything in this file:
//   * the translation unit is compiled with -ffp-contract=off (hipcc contracts by default);
//   * the only fused operation ever used is fma(2.0, zr*zi, ci): 2*p is exact, so it rounds once,
//     exactly like fl(fl((2*zr)*zi) + ci) -- EXCEPT when zr*zi is subnormal.  The host therefore
//     selects the "exact doubling" instantiation (kFmaDouble = false) whenever a non-zero imaginary
//     coordinate is small enough for that to matter (see mbk_api.hip: needs_safe_doubling);
//   * squares are shared between the bailout test of step n and the update of step n+1 (same
//     operands, same operation, same rounding).
//
// Roofline that bounds these kernels: fp64 VALU issue rate (not HBM, not MFMA): 6 fp64 VALU
// operations per pixel-iteration (3 mul, 2 add, 1 fma) plus the bailout test (1 add + 1 v_cmp, each a
// full issue slot on gfx950) -- per step in kernels "simple"/"asm", once per 8 steps in "group";
// algorithmic HBM traffic is the 4 B (int32) and/or 1 B (uint8) written per pixel, nothing is read.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mbk {

// One axis of a view, prepared on the host with individually rounded fp64 operations.
struct Axis {
    double start;  // x[0]
    double step;   // fl(delta / (n-1))
    double last;   // value of sample n-1: stop (n > 1) or start (n == 1)
    double delta;  // fl(stop - start)           (only used by numpy's step == 0 fallback)
    double div;    // (double)(n-1)              (idem)
    uint32_t n;
    uint32_t step_is_zero;  // numpy: y = (k/div)*delta instead of k*step
};

struct ReduceSlot;   // (below: partial results of the statistics reduction)

struct TileArgs {
    Axis re, im;
    uint32_t col0, row0, ncols, nrows;
    uint32_t blocks_x;    // workgroups per row of 8-pixel-high block rows (1-D grid)
    uint32_t out_pitch, out_col0, out_row0;  // output element = (row + out_row0) * out_pitch + col + out_col0
    int32_t mrd;
    uint32_t quant_wide;  // 1: count*256+mrd-1 does not fit 32 bits -> 64-bit quantiser division
    double quant_rcp;     // fl(1/mrd), host-computed: the narrow quantiser divides by multiplying (see quantise)
    uint32_t exact_steps; // steps tested one by one before the grouped test takes over
    uint32_t exact_steps_long;  // the same for the blocks that run 16-step groups (classified as interior)
    uint32_t ring_possible;  // 0 = the host proved that no pixel of the window lies near |c| = 2
    uint32_t fast_bx_end, fast_by_end;  // 8x8 blocks with column index < fast_bx_end and row index < fast_by_end
                          // lie wholly inside the window, hold neither axis' pinned last sample, and both steps
                          // are non-zero: their coordinates are fl(fl(k*step)+start), no per-lane edge handling
    uint32_t perm_mul;    // workgroup order: block = (blockIdx * perm_mul) mod gridDim (1 = row-major)
    const uint32_t *order; // optional dispatch order (three-class list from classify_blocks_kernel, whose comment has
                           // the layout); an entry is (block row << 16) | workgroup column -- the host offers it only
                           // when both fit 16 bits
    uint32_t ngrid;        // with `order`: the grid size (= list length); the counters sit at order[ngrid .. ngrid + 3)
    uint32_t order_mid;    // with `order`: 1 = a middle class was built (MBK_OPT_PROBE_MID <= probe depth), dispatched between
                           // the heavy and the light blocks; 2 = built from MBK_OPT_M_LATE and dispatched FIRST
    uint32_t unit_stride;  // kernel "units" (mbk_units.h): the grid size G; workgroup j takes units j, j + G, ...
    uint32_t stamp_tag;    // kernel "units": 16-bit launch number written into the top of every time stamp
    const uint32_t *plan;  // kernel "units": the shares of the eight XCDs (units_plan; layout in mbk_units.h)
    unsigned long long *stamps;  // kernel "units", may be null: pinned host memory for this launch's time stamps
    int32_t *counts;      // may be null
    uint8_t *bytes;       // may be null
    double *smooth;       // may be null: continuous escape-time value (BASELINE cfg5), see smooth_value
    ReduceSlot *stats;    // may be null; honoured by the finish-in-place light pass only (mbk_scan.h, kStats): the
                          // kernel adds the tile's pixel-iterations and never-escaped count to these partial results
                          // itself, so that a DataChunk needs no int32 counts in HBM at all
    uint32_t cyc_window;  // cycle test: the reference state's window grows by a quarter while shorter than this many checks,
                          // doubles from there on (0 = always doubles; mbk_loops.inc, WINDOW SCHEDULE)
    // SPILL (round 6; block_pixel_spill below, mbk_spill.h): spill_first != 0 = the one-wave-per-block kernel hands the last few
    // live lanes of a block over to a second pass instead of running them alone
    uint32_t spill_first; // steps between the per-step prologue and the first checkpoint (a multiple of 32); each next one twice as far
    uint32_t spill_lanes; // a block spills at a checkpoint when this many lanes or fewer are still alive (slots per block)
    uint32_t spill_win_shift;  // second pass, cycle test: the first window is (steps behind the wave's lanes) >> this, in checks
    void *spill_z;        // [block * spill_lanes + rank]: (zr, zi) of a spilled lane, as two T
    uint32_t *spill_meta; // [block * spill_lanes + rank]: lane in the block (bits 0..5) | steps done (bits 6..31)
    uint32_t *spill_cnt;  // [block]: lanes the block spilled | the number of its checkpoint << 8 (zeroed by the host before the launch)
};

// np.linspace sample k (numpy/_core/function_base.py): two roundings, endpoint pinned.
__device__ __forceinline__ double axis_value(const Axis &a, uint32_t k)
{
    double y;
    if (a.step_is_zero) {
        double q = (double)k / a.div;
        y = q * a.delta;
    } else {
        y = (double)k * a.step;
    }
    double v = y + a.start;
    return (k + 1u == a.n) ? a.last : v;
}

// W.py:96-98 in exact integer form: ceil(count*256/mrd) mod 256 (proved equal to the float form:
// tests/test_oracle.py::test_quantiser_integer_form), i.e. floor(x / mrd) mod 256 with
// x = count*256 + mrd - 1.  The 32-bit integer division costs ~25 VALU instructions per pixel -- a third
// of everything a fast-escaping block executes -- so the narrow path (mrd < 2^23) divides by multiplying:
//     q = trunc(fma((double)x, rcp, 2^-30)),  rcp = fl(1/mrd).
// Exact: count <= mrd-1 gives x/mrd < 2^9, so the fma's total error is below 2^-43; a non-integer
// quotient has a fractional part in [2^-23, 1 - 2^-23], which neither the error nor the 2^-30 bias can
// carry across an integer; an integer quotient k lands in [k + 2^-30 - 2^-43, k + 2^-30 + 2^-43].
// (tests/test_oracle.py::test_quantiser_reciprocal_form restates this with exact rationals.)
__device__ __forceinline__ uint8_t quantise(int32_t count, int32_t mrd, uint32_t wide, double rcp)
{
    if (wide) {
        uint64_t x = (uint64_t)(uint32_t)count * 256ull + (uint64_t)(uint32_t)mrd - 1ull;
        return (uint8_t)(x / (uint64_t)(uint32_t)mrd);
    }
    const uint32_t x = (uint32_t)count * 256u + (uint32_t)mrd - 1u;
    return (uint8_t)(uint32_t)__builtin_fma((double)x, rcp, 0x1p-30);
}
__device__ __forceinline__ uint8_t quantise(int32_t count, const TileArgs &p)
{
    return quantise(count, p.mrd, p.quant_wide, p.quant_rcp);
}

// BASELINE config 5 (NOT in the reference): continuous ("smooth") escape-time value at the
// reference's own bailout, nu = n + 1 - log2(0.5 * ln |z_n|^2) for an escaped pixel (n = escape index,
// |z_n|^2 = the value that tripped `>= 4`), 0 for a pixel that never escaped.  The integer information
// is the exact count; only log/log2 are subject to libm-vs-ocml rounding (tests allow 1e-12).
__device__ __forceinline__ double smooth_value(int32_t count, double m)
{
    return count > 0 ? (double)count + 1.0 - log2(0.5 * log(m)) : 0.0;
}

// The reference loop for one pixel.  kFmaDouble selects how fl(2*zr*zi + ci) is formed.
template <bool kFmaDouble>
__device__ __forceinline__ int32_t escape_count(double cr, double ci, int32_t mrd)
{
    double zr = cr, zi = ci;
    double a = zr * zr, b = zi * zi;
    int32_t result = 0;
    for (int32_t n = 1; n < mrd; ++n) {
        double t = a - b;
        double zi_new;
        if (kFmaDouble) {
            double p = zr * zi;
            zi_new = __builtin_fma(2.0, p, ci);
        } else {
            double w = 2.0 * zr;
            double u = w * zi;
            zi_new = u + ci;
        }
        zr = t + cr;
        zi = zi_new;
        a = zr * zr;
        b = zi * zi;
        double m = a + b;
        if (m >= 4.0) {
            result = n;
            break;
        }
    }
    return result;
}

// ---------------------------------------------------------------------------------------------
// Kernel "simple": one lane per pixel, one 8x8 pixel block per wavefront (compact blocks keep the
// 64 lanes' iteration counts coherent: SURVEY.md P4 lane efficiency 0.90 vs 0.77 for 64x1 strips),
// four wavefronts side by side per 256-thread workgroup (32 x 8 pixels).  The divergent loop exits a
// wavefront as soon as all 64 lanes have escaped (EXEC == 0).
// ---------------------------------------------------------------------------------------------
template <bool kFmaDouble>
__global__ __launch_bounds__(256) void tile_simple_kernel(TileArgs p)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t by = blockIdx.x / p.blocks_x, bx = blockIdx.x - by * p.blocks_x;
    const uint32_t lc = bx * 32u + wave * 8u + (lane & 7u);  // column inside the window
    const uint32_t lr = by * 8u + (lane >> 3);               // row inside the window
    if (lc >= p.ncols || lr >= p.nrows) return;
    const double cr = axis_value(p.re, p.col0 + lc);
    const double ci = axis_value(p.im, p.row0 + lr);
    const int32_t count = escape_count<kFmaDouble>(cr, ci, p.mrd);
    const size_t o = (size_t)lr * p.ncols + lc;
    if (p.counts) p.counts[o] = count;
    if (p.bytes) p.bytes[o] = quantise(count, p);
}

// The hand-scheduled loops (kernels "asm" and "group") live in mbk_loops.inc, instantiated for
// double (the reference's arithmetic) and float (cfg4's fp32 variant) as overloads.
#define MBK_T double
#define MBK_F "f64"
#define MBK_CI_FROM_T64 "v_add_f64 %[ci], %[t64], %[start]\n"
#define MBK_CR_FROM_T64 "v_add_f64 %[cr], %[t64], %[start]\n"
#define MBK_CYC_BITS "u64"   // bitwise state compare / copy of the cycle test
#define MBK_CYC_MOV "b64"
#include "mbk_loops.inc"
#undef MBK_T
#undef MBK_F
#undef MBK_CI_FROM_T64
#undef MBK_CR_FROM_T64
#undef MBK_CYC_BITS
#undef MBK_CYC_MOV
#define MBK_T float
#define MBK_F "f32"
#define MBK_CI_FROM_T64 "v_add_f64 %[t64], %[t64], %[start]\nv_cvt_f32_f64 %[ci], %[t64]\n"
#define MBK_CR_FROM_T64 "v_add_f64 %[t64], %[t64], %[start]\nv_cvt_f32_f64 %[cr], %[t64]\n"
#define MBK_CYC_BITS "u32"
#define MBK_CYC_MOV "b32"
#include "mbk_loops.inc"
#undef MBK_T
#undef MBK_F
#undef MBK_CI_FROM_T64
#undef MBK_CR_FROM_T64
#undef MBK_CYC_BITS
#undef MBK_CYC_MOV


// One pixel of an 8x8 block, start to finish (one lane each; the whole wave calls it): coordinates, the
// escape loop of the chosen kind, the stores.  T = double: the reference's arithmetic.  T = float: the fp32
// variant (coordinates are generated in fp64 exactly as for the fp64 path and then rounded once to fp32; the
// loop is strict fp32).  long_groups (wave-uniform, kGroup == 16 only): 16-step groups for this block.
// interior (wave-uniform): the block lies inside TileArgs::fast_bx_end / fast_by_end -- no lane is outside the
// window or on an axis' pinned end point, so the coordinates need no per-lane selects, and when the host has
// also ruled out the |c| = 2 ring for the whole window the per-wave ring test goes too (~20 of the ~57 VALU
// instructions a block costs outside its loop; on cfg2 that overhead is 15 M of 317 M instructions).
// kCycle: the grouped loops retire exactly periodic orbits early (mbk_loops.inc, MBK_G_CYC).
// (ucol, urow): the block's first column / row inside the window, wave-uniform; (lx, ly): this lane's pixel in the block.
// Returns the lane's count (-1: the lane has no pixel -- outside the window).
template <typename T, bool kFmaDouble, int kGroup, bool kCycle = false>
__device__ __forceinline__ int32_t block_pixel(const TileArgs &p, uint32_t ucol, uint32_t urow, uint32_t lx, uint32_t ly,
                                               bool long_groups, bool interior = false)
{
    const uint32_t lc = ucol + lx, lr = urow + ly;
    T cr, ci;
    bool ring_test = kGroup != 0 && kFmaDouble;
    if (interior) {
        cr = (T)((double)(p.col0 + lc) * p.re.step + p.re.start);
        ci = (T)((double)(p.row0 + lr) * p.im.step + p.im.start);
        ring_test = ring_test && p.ring_possible != 0u;
    } else {
        if (lc >= p.ncols || lr >= p.nrows) return -1;
        cr = (T)axis_value(p.re, p.col0 + lc);
        ci = (T)axis_value(p.im, p.row0 + lr);
    }
    int32_t count;
    T m = 0;  // |z|^2 at the escaping step (only meaningful when count > 0)
    if (kGroup != 0 && kFmaDouble) {
        // the grouped test relies on "|z|^2 >= 4 stays >= 4"; only |c| within rounding of 2 could
        // spoil that, so any wave touching that ring takes the per-step loop (wave-uniform branch)
        bool risky = false;
        if (ring_test) {
            const T c2 = cr * cr + ci * ci;
            const T margin = sizeof(T) == 8 ? (T)1e-9 : (T)1e-3;
            // (c2 - 4 is exact near 4 -- Sterbenz --, so this is the band 4 +- margin in two instructions)
            risky = __ballot(__builtin_fabs((double)(c2 - (T)4)) < (double)margin) != 0ull;
        }
        if (risky) {
            count = escape_count_asm<true>(cr, ci, p.mrd, &m);
        } else if (kGroup == 32) {
            // 32-step groups (6.0625 slots per step; strict loops only) for the blocks classified as interior, 8 elsewhere
            count = long_groups ? escape_count_group<32, false>(cr, ci, p.mrd, &m, p.exact_steps_long)
                                : escape_count_group<8, false>(cr, ci, p.mrd, &m, p.exact_steps);
        } else if (kGroup == 16) {
            // 16-step groups (6.125 issue slots per step) only where the test hardly ever trips (interior of
            // the set: the blocks a probe or the light pass classified as such); 8 elsewhere
            // (blocks classified as interior skip the per-step prologue when MBK_OPT_EXACT_LONG says so: with the
            // deferred replay a lane that does escape early costs one trip + the block's single fix-up)
            count = long_groups ? escape_count_group<16, kCycle>(cr, ci, p.mrd, &m, p.exact_steps_long, p.cyc_window)
                                : escape_count_group<8, kCycle>(cr, ci, p.mrd, &m, p.exact_steps, p.cyc_window);
        } else {
            count = escape_count_group<kGroup == 8 ? 8 : 4, kCycle>(cr, ci, p.mrd, &m, p.exact_steps, p.cyc_window);
        }
    } else {
        count = escape_count_asm<kFmaDouble>(cr, ci, p.mrd, &m);
    }
    // output element = scalar base of the block (64-bit, on the scalar unit) + a 32-bit lane offset: ly * pitch < 2^31
    // because the window has more than ly rows and at most 2^31 pixels (validate_view)
    const size_t ubase = (size_t)(urow + p.out_row0) * p.out_pitch + ucol + p.out_col0;
    const uint32_t loff = ly * p.out_pitch + lx;
    if (p.counts) (p.counts + ubase)[loff] = count;
    if (p.bytes) (p.bytes + ubase)[loff] = quantise(count, p);
    if (p.smooth) (p.smooth + ubase)[loff] = smooth_value(count, (double)m);
    return count;
}

// ---------------------------------------------------------------------------------------------
// SPILL (round 6).  A wave runs until its LAST lane is done.  On a deep zoom most blocks hold pixels that escape after a few
// hundred steps next to a handful that need thousands (filaments) or never escape: cfg3 executes 1 219.8 M wave-steps for
// 1 017.9 M wave-steps of work (lane activity 0.835, measured 0.832), and 313 000 of its 1 048 576 blocks reach step 512 with
// 16 or fewer of their 64 lanes alive.  Re-packing the survivors inside a workgroup loses (round 4: 16x16 regions reach 0.85
// and pay the 4-wave-workgroup penalty); a lane-refill kernel loses (round 5: its grouped test replays every group).  What is
// left is to pack them GLOBALLY: at checkpoints -- spill_first steps after the per-step prologue, then twice as far each time --
// a block with at most spill_lanes live lanes writes their state (zr, zi, lane, steps done) into its own slots of a list in
// HBM and ends; a prefix sum over the blocks' counts compacts the list (no atomics: a device-scope atomic with return costs
// 25 ns, serialised chip-wide -- profiles/r04/units_pool_ab.txt -- and a cfg3 launch would need 380 000 of them), and a second
// kernel (tile_spill_kernel, mbk_spill.h) runs the listed lanes 64 to a wave from where they stopped: escape_steps_tail resumes
// from any state, and the deferred replay keeps the grouped test cheap there (a lane that trips just leaves).  Model on the
// exact counts (scripts/spill_model.py): cfg3 1 219.8 -> 1 106.4 + 23.3 M wave-steps (-7.4 %) with checkpoints 256, 512, ... and
// 16 lanes.  Exact by construction: the same recurrence from the same state; the cycle test keeps ONE schedule across a block's
// stretches (escape_steps_carry) and starts afresh in the second pass (any schedule of its reference state is exact).  Interior
// blocks only (regular coordinates, no lane outside the window).
// The caller stores nothing for a spilled lane (the second pass does).
// ---------------------------------------------------------------------------------------------
template <typename T> struct SpillPair;
template <> struct SpillPair<double> { using type = double2; };
template <> struct SpillPair<float> { using type = float2; };
constexpr int32_t kSpilled = -2;   // count of a lane whose pixel the second pass finishes

template <typename T, int kGroup, bool kCycle>
__device__ __forceinline__ void block_pixel_spill(const TileArgs &p, uint32_t ucol, uint32_t urow, uint32_t lx, uint32_t ly,
                                                  bool long_groups, uint32_t block_index)
{
    const uint32_t lc = ucol + lx, lr = urow + ly;
    const T cr = (T)((double)(p.col0 + lc) * p.re.step + p.re.start);
    const T ci = (T)((double)(p.row0 + lr) * p.im.step + p.im.start);
    bool risky = false;
    if (p.ring_possible != 0u) {   // (block_pixel: a wave touching the |c| = 2 ring takes the per-step loop)
        const T c2 = cr * cr + ci * ci;
        const T margin = sizeof(T) == 8 ? (T)1e-9 : (T)1e-3;
        risky = __ballot(__builtin_fabs((double)(c2 - (T)4)) < (double)margin) != 0ull;
    }
    int32_t count;
    if (risky) {
        count = escape_count_asm<true>(cr, ci, p.mrd);
    } else {
        T zr = cr, zi = ci, a = zr * zr, b = zi * zi, m = 0;
        int32_t cnt = 0;
        const uint32_t total = p.mrd > 1 ? (uint32_t)p.mrd - 1u : 0u;
        const uint32_t ex = (kGroup >= 16 && long_groups) ? p.exact_steps_long : p.exact_steps;
        const uint32_t first = total < ex ? total : ex;
        escape_steps_asm<true>(cr, ci, zr, zi, a, b, m, cnt, 0u, first);
        uint32_t n = first, seg = p.spill_first, level = 0u;     // wave-uniform
        bool spilled = false;
        // the cycle test's reference state, check counter and window, carried across the stretches (escape_steps_carry)
        T szr = zr, szi = zi;
        uint32_t cyc_tc = 0u, cyc_win = 1u;
        // The stretches run WITHOUT the deferred replay at their ends: a lane that tripped a group test stays pending -- frozen,
        // outside every later stretch (cnt != 0) -- and ONE fix-up behind the loop serves the whole block, as in block_pixel
        // (the first build replayed at every checkpoint: up to 16 exact steps x 9 slots per stretch in which a lane escaped,
        // which on a deep zoom is every stretch of every block -- it cost what the spill saved, profiles/r06/spill_ab.txt).
        while (n < total) {
            if (__ballot(cnt == 0) == 0ull) break;   // every lane has escaped or been retired
            // the next checkpoint -- unless fewer than 64 steps would be left behind it: then straight to the end
            const uint32_t stop = (total - n > seg + 64u) ? n + seg : total;
            if (cnt == 0) {
                if (kCycle) {
                    if (kGroup >= 16 && long_groups) escape_steps_carry<16>(cr, ci, zr, zi, a, b, m, cnt, n, stop, p.cyc_window, szr, szi, cyc_tc, cyc_win);
                    else escape_steps_carry<8>(cr, ci, zr, zi, a, b, m, cnt, n, stop, p.cyc_window, szr, szi, cyc_tc, cyc_win);
                } else {
                    if (kGroup >= 16 && long_groups) escape_steps_tail<16, false, false>(cr, ci, zr, zi, a, b, m, cnt, n, stop);
                    else escape_steps_tail<8, false, false>(cr, ci, zr, zi, a, b, m, cnt, n, stop);
                }
            }
            if (kCycle) {   // (uniform values that came out of a divergent region: say so)
                cyc_tc = (uint32_t)__builtin_amdgcn_readfirstlane((int)cyc_tc);
                cyc_win = (uint32_t)__builtin_amdgcn_readfirstlane((int)cyc_win);
            }
            n = stop;
            if (n >= total) break;
            const unsigned long long alive = __ballot(cnt == 0);
            const uint32_t k = (uint32_t)__popcll(alive);
            if (k == 0u) break;
            if (k <= p.spill_lanes) {
                if (cnt == 0) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(alive >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)alive, 0u));
                    const size_t slot = (size_t)block_index * p.spill_lanes + rank;
                    typename SpillPair<T>::type z;
                    z.x = zr;
                    z.y = zi;
                    reinterpret_cast<typename SpillPair<T>::type *>(p.spill_z)[slot] = z;
                    p.spill_meta[slot] = (ly * 8u + lx) | (n << 6);
                    if (rank == 0u) p.spill_cnt[block_index] = k | ((level < 11u ? level : 11u) << 8);   // (11 = kSpillLevels - 1)
                    spilled = true;
                }
                break;
            }
            // the next checkpoint lies twice as far behind the prologue (3/2 and 4/3 in turn -- 256, 384, 512, 768, ... -- was
            // tried: strict +-0, cycle leg +6 %: every checkpoint restarts the cycle test's window; profiles/r06/spill_ab.txt)
            seg = n - first;
            ++level;
        }
        escape_fixup(cr, ci, zr, zi, a, b, m, cnt);   // (a spilled lane has cnt == 0: not pending)
        count = spilled ? kSpilled : ((kCycle && cnt == -1) ? 0 : cnt);   // -1: retired by the cycle test = never escapes
    }
    if (count != kSpilled) {
        const size_t ubase = (size_t)(urow + p.out_row0) * p.out_pitch + ucol + p.out_col0;
        const uint32_t loff = ly * p.out_pitch + lx;
        if (p.counts) (p.counts + ubase)[loff] = count;
        if (p.bytes) (p.bytes + ubase)[loff] = quantise(count, p);
    }
}

// Kernels "asm" (kGroup = 0) and "group": one 8x8 block per wave, blockDim / 64 blocks per workgroup.
// kSpill (single-wave workgroups, kFmaDouble, no smooth output): interior blocks go through block_pixel_spill.
template <typename T, bool kFmaDouble, int kGroup = 0, bool kCycle = false, bool kSpill = false>
__global__ __launch_bounds__(256) void tile_asm_kernel(TileArgs p)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // Dispatch order != image order when an order list is given (heavy-first, classify_blocks_kernel)
    // or perm_mul != 1 (multiplicative permutation, coprime to the grid size).
    uint32_t bx, by;
    uint32_t n_heavy = 0u, heavy_lo = 0u;
    if (p.order) {
        // classes (classify_blocks_kernel): heavy at the front of the list, light filled in from its back, and -- only
        // when a middle class was built (p.order_mid) -- that class in a list of its own right behind the three
        // counters.  Without it a wave needs ONE load for its entry, issued together with the heavy count: the grid
        // size comes with the arguments (p.ngrid), not from the dispatch packet, so a wave's start is two scalar
        // round trips (arguments, then entry + count) instead of four -- a light block lives for little else.
        const uint32_t j = blockIdx.x;
        uint32_t e;
        if (p.order_mid == 2u) {
            // round 5, "M late first" (mbk_units.h has the why): the middle class -- boundary blocks whose centre escapes
            // late, among them the few that hold a never-escaping pixel and run as long as an interior block -- opens the
            // dispatch order instead of sitting between the heavy and the light blocks
            n_heavy = p.order[p.ngrid];
            const uint32_t n_mid = p.order[p.ngrid + 2u];
            e = j < n_mid ? p.order[p.ngrid + 3u + j] : (j < n_mid + n_heavy ? p.order[j - n_mid] : p.order[j]);
            heavy_lo = n_mid;
        } else if (p.order_mid) {
            n_heavy = p.order[p.ngrid];
            const uint32_t n_mid = p.order[p.ngrid + 2u];
            e = (j >= n_heavy && j < n_heavy + n_mid) ? p.order[p.ngrid + 3u + (j - n_heavy)] : p.order[j];
        } else {
            e = p.order[j];   // packed: no division here
            n_heavy = p.order[p.ngrid];
        }
        by = e >> 16;
        bx = e & 0xffffu;
    } else {
        const uint32_t blk = (uint32_t)(((uint64_t)blockIdx.x * p.perm_mul) % gridDim.x);
        by = blk / p.blocks_x;
        bx = blk - by * p.blocks_x;
    }
    const uint32_t wcol = bx * (blockDim.x >> 6) + wave;  // one 8x8 block per wave
    // the blocks the heavy-first probe put at the front of the dispatch order take the 16-step groups
    const bool long_groups = kGroup >= 16 && (!p.order || (blockIdx.x >= heavy_lo && blockIdx.x < heavy_lo + n_heavy));   // wave-uniform
    const bool interior = wcol < p.fast_bx_end && by < p.fast_by_end;                        // wave-uniform
    if (kSpill && kFmaDouble && kGroup >= 8 && interior)
        block_pixel_spill<T, kGroup, kCycle>(p, wcol * 8u, by * 8u, lane & 7u, lane >> 3, long_groups, by * p.blocks_x + wcol);
    else
    block_pixel<T, kFmaDouble, kGroup, kCycle>(p, wcol * 8u, by * 8u, lane & 7u, lane >> 3, long_groups, interior);
}

// ---------------------------------------------------------------------------------------------
// Dispatch-order pre-pass ("heavy first").  Why: the hardware hands workgroups out in index order.
// In image order, cheap blocks (every pixel escapes within a few steps: 83 % of cfg2) are interleaved
// with in-set blocks; each time a long-running wave retires, its slot hosts a string of cheap waves
// (~10 us each, almost no VALU work) before the next long one arrives, so the SIMDs run with ~5
// useful waves instead of 8, and the tile ends with a drain of long waves and nothing to overlap it.
// Measured on cfg2 (profiles/microbench/occupancy_trace.hip): heavy-first order -10 % kernel time.
// This kernel probes ONE pixel per workgroup region (its centre) for `probe_steps` steps and builds the
// order in three classes (round 3; two before): "heavy" = the probe did not escape -> front of `order` (atomic
// cursor counters[0]); "light" = the probe escaped within `mid_min` - 1 steps -> filled in from the back
// (counters[1]); "middle" = the probe escaped later (count >= mid_min) -> a list of its own behind the counters
// (counters[2]), dispatched between the two.  Why the middle class: scripts/dispatch_model.py (event simulation on
// the tile's exact block durations) puts the idle share of a cfg2 launch not in the order of the in-set blocks
// but in the ~20 000 boundary blocks whose centre pixel escapes early -- filed under "light", they were
// dispatched last and formed the tail (makespan 1.040x the balanced ideal; 1.018x with this order).
// It is a scheduling heuristic only: a mis-classified block is merely computed earlier or later; results
// cannot change.  Layout: order[0 .. n) heavy from the front / light from the back, order[n .. n+3) the three
// counters (heavy, light, middle), order[n+3 .. 2n+3) the middle list.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void classify_blocks_kernel(TileArgs p, uint32_t nregions,
                                                                uint32_t region_w, int32_t probe_steps, int32_t mid_min,
                                                                uint32_t *order, uint32_t *counters)
{
    __shared__ uint32_t s_cnt[3][16], s_base[3];
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool valid = r < nregions;
    uint32_t cls = 1u;     // 0 heavy, 1 light, 2 middle
    uint32_t packed = 0;   // the list entry of this region: (block row << 16) | workgroup column
    if (valid) {
        const uint32_t by = r / p.blocks_x, bx = r - by * p.blocks_x;
        packed = (by << 16) | bx;
        uint32_t lc = bx * region_w + region_w / 2u, lr = by * 8u + 4u;
        lc = lc < p.ncols ? lc : p.ncols - 1u;
        lr = lr < p.nrows ? lr : p.nrows - 1u;
        const double cr = axis_value(p.re, p.col0 + lc), ci = axis_value(p.im, p.row0 + lr);
        const int32_t cap = p.mrd < probe_steps ? p.mrd : probe_steps;
        const int32_t cnt = cap > 1 ? escape_count<true>(cr, ci, cap) : 1;
        cls = cnt == 0 ? 0u : (cnt >= mid_min ? 2u : 1u);
    }
    unsigned long long m[3];
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) {
        m[k] = __ballot(valid && cls == k);
        if (lane == 0) s_cnt[k][wave] = (uint32_t)__popcll(m[k]);
    }
    __syncthreads();
    if (threadIdx.x < 3u) {
        const uint32_t k = threadIdx.x, nw = (blockDim.x + 63u) >> 6;
        uint32_t t = 0;
        for (uint32_t w = 0; w < nw; ++w) {
            const uint32_t c = s_cnt[k][w];
            s_cnt[k][w] = t;
            t += c;
        }
        s_base[k] = t ? atomicAdd(&counters[k], t) : 0u;
    }
    __syncthreads();
    if (valid) {
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t i = s_base[cls] + s_cnt[cls][wave] + (uint32_t)__popcll(m[cls] & below);
        if (cls == 0u) order[i] = packed;
        else if (cls == 1u) order[nregions - 1u - i] = packed;
        else order[nregions + 3u + i] = packed;
    }
}

// ---------------------------------------------------------------------------------------------
// Reduction over finished results: pixel-iterations, never-escaped pixels, all-zero / all-one byte
// flags (DataChunk.cs:82,87).  HBM-bound, 4-5 B/pixel read once; not part of the timed hot loop.
// ---------------------------------------------------------------------------------------------
struct ReduceOut {
    unsigned long long pixel_iterations;
    unsigned long long never_pixels;
    unsigned long long run_starts;  // positions i with i == 0 or bytes[i] != bytes[i-1]
    unsigned int any_byte_not_zero;
    unsigned int any_byte_not_one;
};
// The kernels accumulate into kReduceSlots partial results, each on its own 128-byte line, and the host adds
// them up: with one result record every wave's atomics went to the same line, and same-address atomics are
// serialised by the L2 (8 192 waves x 3 atomics: the reduction of a 64 MiB tile took 166 us, 0.4 TB/s, whatever
// the loads looked like).
constexpr uint32_t kReduceSlots = 64;
struct alignas(128) ReduceSlot {
    ReduceOut r;
};

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void reduce_kernel(const int32_t *__restrict__ counts,
                                                     const uint8_t *__restrict__ bytes,
                                                     uint64_t n, uint32_t mrd, ReduceSlot *slots)
{
    ReduceOut *out = &slots[blockIdx.x % kReduceSlots].r;
    const unsigned long long cap = mrd > 1u ? (unsigned long long)mrd - 1ull : 0ull;
    unsigned long long iters = 0, never = 0, starts = 0;
    unsigned int nz = 0, no = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (counts) {
            int32_t c = counts[i];
            iters += c > 0 ? (unsigned long long)c : cap;
            never += c == 0 ? 1ull : 0ull;
        }
        if (bytes) {
            uint8_t b = bytes[i];
            nz |= (b != 0);
            no |= (b != 1);
            starts += (i == 0 || bytes[i - 1] != b) ? 1ull : 0ull;
        }
    }
    iters = wave_sum_u64(iters);
    never = wave_sum_u64(never);
    starts = wave_sum_u64(starts);
    const unsigned long long nzb = __ballot(nz != 0), nob = __ballot(no != 0);
    if ((threadIdx.x & 63u) == 0) {
        if (iters) atomicAdd(&out->pixel_iterations, iters);
        if (never) atomicAdd(&out->never_pixels, never);
        if (starts) atomicAdd(&out->run_starts, starts);
        if (nzb) atomicOr(&out->any_byte_not_zero, 1u);
        if (nob) atomicOr(&out->any_byte_not_one, 1u);
    }
}

// The same reduction at HBM speed for the common case (counts 16-byte aligned, bytes 4-byte aligned): four
// pixels per lane and trip -- one 16-byte load of counts, one 4-byte load of bytes, consecutive lanes on
// consecutive groups, so a wave instruction covers 1 KiB / 256 B contiguous.  The scalar kernel above moves one
// byte per lane and load (111 us for a 64 + 16 MiB tile, 0.7 TB/s: three times the kernel time of a light tile).
// Run starts inside a 4-byte word: x = w ^ (w << 8 | previous byte) has a non-zero byte k where byte k differs
// from its predecessor; the byte before the word is loaded separately (same cache line as the neighbour's word).
template <bool kCounts, bool kBytes>
__global__ __launch_bounds__(256) void reduce_vec_kernel(const int32_t *__restrict__ counts,
                                                         const uint8_t *__restrict__ bytes,
                                                         uint64_t n, uint32_t mrd, ReduceSlot *slots)
{
    ReduceOut *out = &slots[blockIdx.x % kReduceSlots].r;
    const unsigned long long cap = mrd > 1u ? (unsigned long long)mrd - 1ull : 0ull;
    unsigned long long iters = 0, never = 0, starts = 0;
    unsigned int nz = 0, no = 0;
    const uint64_t nq = n >> 2;  // whole groups of four pixels (>= 1: the host sends n >= 1024 here)
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    constexpr int kUnroll = 4;   // four independent groups per trip, loaded unconditionally (index clamped) so that
                                 // all loads of a trip are in flight before the first is used
    for (uint64_t q0 = tid; q0 < nq; q0 += kUnroll * stride) {
        int4 c[kUnroll];
        uint32_t w[kUnroll], prev[kUnroll];
#pragma unroll
        for (int k = 0; k < kUnroll; ++k) {
            const uint64_t q = q0 + (uint64_t)k * stride, qc = q < nq ? q : nq - 1u;
            if (kCounts) c[k] = reinterpret_cast<const int4 *>(counts)[qc];
            if (kBytes) {
                w[k] = reinterpret_cast<const uint32_t *>(bytes)[qc];
                prev[k] = bytes[qc ? 4u * qc - 1u : 0u];
            }
        }
#pragma unroll
        for (int k = 0; k < kUnroll; ++k) {
            const uint64_t q = q0 + (uint64_t)k * stride;
            const bool in = q < nq;
            if (kCounts) {
                const unsigned long long it = (c[k].x > 0 ? (unsigned long long)c[k].x : cap) + (c[k].y > 0 ? (unsigned long long)c[k].y : cap) +
                                              (c[k].z > 0 ? (unsigned long long)c[k].z : cap) + (c[k].w > 0 ? (unsigned long long)c[k].w : cap);
                const unsigned int nv = (c[k].x == 0) + (c[k].y == 0) + (c[k].z == 0) + (c[k].w == 0);
                iters += in ? it : 0ull;
                never += in ? nv : 0u;
            }
            if (kBytes) {
                // pixel 0 always starts a run: pretend its predecessor differs
                const uint32_t pb = q ? prev[k] : (~w[k] & 0xffu);
                const uint32_t x = w[k] ^ ((w[k] << 8) | pb);
                const unsigned int st = ((x & 0xffu) != 0u) + ((x & 0xff00u) != 0u) + ((x & 0xff0000u) != 0u) + ((x & 0xff000000u) != 0u);
                starts += in ? st : 0u;
                nz |= (in && w[k] != 0u);
                no |= (in && w[k] != 0x01010101u);
            }
        }
    }
    const uint64_t i = (nq << 2) + tid;  // the last n mod 4 pixels, one lane each
    if (i < n) {
        if (kCounts) {
            const int32_t c = counts[i];
            iters += c > 0 ? (unsigned long long)c : cap;
            never += c == 0 ? 1ull : 0ull;
        }
        if (kBytes) {
            const uint8_t b = bytes[i];
            nz |= (b != 0);
            no |= (b != 1);
            starts += (i == 0 || bytes[i - 1] != b) ? 1ull : 0ull;
        }
    }
    iters = wave_sum_u64(iters);
    never = wave_sum_u64(never);
    starts = wave_sum_u64(starts);
    const unsigned long long nzb = __ballot(nz != 0), nob = __ballot(no != 0);
    if ((threadIdx.x & 63u) == 0) {
        if (iters) atomicAdd(&out->pixel_iterations, iters);
        if (never) atomicAdd(&out->never_pixels, never);
        if (starts) atomicAdd(&out->run_starts, starts);
        if (nzb) atomicOr(&out->any_byte_not_zero, 1u);
        if (nob) atomicOr(&out->any_byte_not_one, 1u);
    }
}

// ---------------------------------------------------------------------------------------------
// On-device DataChunk RLE serialiser (DataChunkSerializer.cs:56-100): run starts -> block counts ->
// scan -> (start, value) per run -> 5-byte records.  HBM-bound: the 16 MiB tile is read twice.
// One byte per thread, 1024-thread workgroups; run starts are found with ballots.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kRleBlock = 1024;

__device__ __forceinline__ bool rle_is_start(const uint8_t *__restrict__ bytes, uint64_t i, uint64_t n)
{
    return i < n && (i == 0 || bytes[i] != bytes[i - 1]);
}

__global__ __launch_bounds__(1024) void rle_count_kernel(const uint8_t *__restrict__ bytes, uint64_t n,
                                                         uint32_t *block_counts)
{
    __shared__ uint32_t s_cnt[16];
    const uint64_t i = (uint64_t)blockIdx.x * kRleBlock + threadIdx.x;
    const unsigned long long m = __ballot(rle_is_start(bytes, i, n));
    if ((threadIdx.x & 63u) == 0) s_cnt[threadIdx.x >> 6] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < 16; ++w) t += s_cnt[w];
        block_counts[blockIdx.x] = t;
    }
}

// single workgroup: exclusive scan of the block counts in place; total -> *total_runs
__global__ __launch_bounds__(1024) void rle_scan_kernel(uint32_t *block_counts, uint32_t nblocks,
                                                        unsigned long long *total_runs)
{
    __shared__ unsigned long long s_part[1024];
    const uint32_t per = (nblocks + 1023u) / 1024u;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < nblocks ? lo + per : nblocks;
    unsigned long long sum = 0;
    for (uint32_t k = lo; k < hi; ++k) sum += block_counts[k];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int t = 0; t < 1024; ++t) {
            const unsigned long long v = s_part[t];
            s_part[t] = run;
            run += v;
        }
        *total_runs = run;
    }
    __syncthreads();
    unsigned long long run = s_part[threadIdx.x];
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t v = block_counts[k];
        block_counts[k] = (uint32_t)run;  // a tile has < 2^32 runs
        run += v;
    }
}

__global__ __launch_bounds__(1024) void rle_scatter_kernel(const uint8_t *__restrict__ bytes, uint64_t n,
                                                           const uint32_t *__restrict__ block_offsets,
                                                           uint32_t *run_start, uint8_t *run_value)
{
    __shared__ uint32_t s_off[16];
    const uint64_t i = (uint64_t)blockIdx.x * kRleBlock + threadIdx.x;
    const bool st = rle_is_start(bytes, i, n);
    const unsigned long long m = __ballot(st);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) s_off[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = block_offsets[blockIdx.x];
        for (int w = 0; w < 16; ++w) {
            const uint32_t v = s_off[w];
            s_off[w] = run;
            run += v;
        }
    }
    __syncthreads();
    if (st) {
        const uint32_t r = s_off[wave] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        run_start[r] = (uint32_t)i;
        run_value[r] = bytes[i];
    }
}

// record r at out + 1 + 5 r: u32 runLength (little-endian), u8 value; out[0] = codec byte
__global__ __launch_bounds__(256) void rle_emit_kernel(const uint32_t *__restrict__ run_start,
                                                       const uint8_t *__restrict__ run_value,
                                                       uint64_t runs, uint64_t n, uint8_t *out)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) out[0] = 0x01;
    if (r >= runs) return;
    const uint32_t len = (uint32_t)((r + 1 < runs ? (uint64_t)run_start[r + 1] : n) - run_start[r]);
    uint8_t *o = out + 1 + 5 * r;
    o[0] = (uint8_t)len;
    o[1] = (uint8_t)(len >> 8);
    o[2] = (uint8_t)(len >> 16);
    o[3] = (uint8_t)(len >> 24);
    o[4] = run_value[r];
}

}  // namespace mbk

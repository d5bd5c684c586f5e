// mbk_units.h -- kernel "group" with UNITS (round 4): the dispatch order of the one-wave-per-block kernel, rebuilt on what
// round 4 measured about the chip.
//
// Facts (profiles/r04/NOTES.md; valu_issue.txt, the wave_limit sweep, scripts/fifo_model.py):
//   * a SIMD's arbiter serves its OLDEST wave first, and two or three waves saturate the fp64 pipe at 4.04-4.06 shader
//     cycles per instruction -- there is no issue gap to recover inside the loops;
//   * the chip has ONE workgroup dispatcher, 0.28 ns per single-wave workgroup, in list order, and a block that is gone
//     after a few steps lives ~1 us (launch, two scalar round trips, ~60 vector instructions of set-up, stores);
//   * so a cfg2 launch (262 144 blocks, 167 000 of them gone within 3 steps) ends with a ~50 us phase in which the
//     dispatcher deals out light blocks one by one while the last heavy waves drain: 8 % of the launch, 12 % on the
//     DataChunk (1,0,0) tile.  An event model with those three facts reproduces the measured launches (cfg2 544.5 vs 547 us,
//     chunk_l1 335.7 vs 336 us, round 3's three-class experiment +0.9 % vs +0.7 %) and predicts -5.6 % / -11.5 % for the
//     order built here.
//
// Three classes from the one-pixel probe of classify (centre pixel of every 8x8 block, 32 steps):
//   H  the probe did not escape        -> one workgroup per block, 16-step groups          (front of the list)
//   M  it escaped at step 4..31, or the block touches a ragged edge / a pinned end point
//                                      -> one workgroup per block, 8-step groups           (list of its own)
//   V  it escaped within 3 steps       -> ROW UNITS: the V blocks of an aligned group of 8 block columns of one block row
//                                         are ONE workgroup, which runs them through the light path (four steps of the
//                                         reference loop with the reference's own test, mbk_loops.inc: escape_light_row)
//                                         and finishes in place, with the code of H/M, whatever block that path cannot.
// Dispatch order H, M, V: the longest jobs first, the boundary blocks next (they used to be filed under "light" and
// dispatched last: round 3's middle class pulled them forward and LOST, because it left 200 000 single light blocks as
// a dispatch-bound tail -- here that tail is 21 000 units), the units last, as the filler of the drain.
// The host cannot know how many units the probe will produce, so the grid is an estimate from its own 256-pixel probe of
// the window and workgroup j takes units j, j + G, j + 2G ... : too small a grid costs a second trip for some workgroups,
// too large a few that leave after one load.  A scheduling heuristic throughout: which class a block lands in changes
// when it is computed and by which of two exact routines, never what is stored.
#pragma once

#include "mbk_refill.h"  // uniform_u32/u64 (includes mbk_kernels.h)

namespace mbk {

// Layout of the list (same buffer as classify_blocks_kernel's): order[0 .. n) H entries from the front and V units from
// the back, order[n .. n+3) the counters (H, V units, M), order[n+3 .. 2n+3) the M entries.  H / M entry: (block row << 16)
// | block column.  V unit: (block row << 16) | (first block column / 8) << 8 | mask of the V blocks among its 8 columns.
// Needs blocks_x % 8 == 0 (a unit never wraps a row), blocks_x <= 2048, block rows < 65536 (the host checks).
// With m_late > 0 the "late" M entries sit at the back of the M region (order[2n+2] downwards, count at order[2n+3]); with
// settle_thr > 0 the settled H entries have a region of their own (order[units_settled_base(n) ..), count at order[2n+4]).
// The dispatch order is: late M, H (unsettled), settled H | M, V units -- the part before the bar is the "front list" that the
// XCD shares deal unevenly.
__host__ __device__ inline size_t units_settled_base(size_t n) { return 2u * n + 64u; }   // (the plan sits in between)
__host__ __device__ inline size_t units_list_words(size_t n) { return 3u * n + 64u; }

//
// M LATE (round 5).  A block with a pixel that never escapes runs all mrd - 1 steps, and from its start to its end it needs
// >= 6 125 dependent-issue instructions ~ 21 us however empty the chip is (a lone wave issues one fp64 instruction per ~8
// cycles: profiles/r04/valu_issue.txt).  The H class is dispatched first, so its long blocks have the whole launch to finish;
// but ~200 blocks of the M class (centre gone within 32 steps, a corner of the block inside the set: cfg2 209 of 47 805,
// DataChunk (1,0,0) 175 of 26 812) run just as long and were dispatched in image order somewhere in the M phase, the last of
// them a few microseconds before the dispatchers ran dry: the launch then ended 15-25 us after its dispatchers had (6-8 % of
// a cycle-test launch, profiles/NOTES.md round 4, 2c).  Every one of those blocks has a centre pixel that escapes at step
// >= 8 (exact counts: 209 of 209, 175 of 175; >= 12 holds 207 / 173), so the probe sorts them out for free: M entries whose
// centre escapes at step >= m_late are filed apart and dispatched FIRST, before H; the rest of M and the V units stay behind
// H as the filler of its drain.
//
struct XcdShares { uint32_t cum[8]; };   // front list: share of XCDs 0..x, in 2^-24 (cum[7] = 2^24)
__host__ __device__ inline void units_plan(uint32_t n_h, uint32_t n_v, uint32_t n_m, const uint32_t *cum, uint32_t stamps, uint32_t *plan,
                                           uint32_t n_ml, uint32_t n_hs);

// H SETTLED (round 5, with the cycle test only).  With the cycle test the H class is no longer uniform: a block whose orbits
// have settled on their attracting cycle retires within a few checks (cfg2: 27 669 of 47 683 H blocks, 194 steps on
// average), one whose orbits have not runs (nearly) all mrd - 1 steps (20 014 blocks, 772 on average, 98.5 % of those that run
// >= 900).  H is dispatched in image order, so the last H blocks to start -- at 8 waves per SIMD the youngest wave of a SIMD
// gets what its elders leave: 50-65 us for a long block -- ended the launch once the late M blocks were out of the way
// (profiles/r05/units_trace_*.txt).  The probe sorts them at no cost: delta = min over p in {1..6, 8} of
// |z_last - z_(last-p)|^2 of the centre pixel after its 32 steps; H blocks with delta <= settle_thr are filed apart and
// dispatched behind the unsettled ones.  (On its own -- without M late -- this order LOST 1.8 % on cfg2: profiles/r05/README.)
__device__ __forceinline__ int32_t probe_settling(double cr, double ci, int32_t cap, double *delta)
{
    double zr = cr, zi = ci;
    int32_t n = 1;
    for (; n < cap - 9; ++n) {                 // the plain loop; the history matters for the last nine states only
        const double t = zr * zr - zi * zi;
        zi = __builtin_fma(2.0, zr * zi, ci);
        zr = t + cr;
        if (zr * zr + zi * zi >= 4.0) { *delta = 1e300; return n; }
    }
    double hr[9], hi[9];                       // hr[p] = z_(last - p) once the loop is through
#pragma unroll
    for (int k = 0; k < 9; ++k) hr[k] = hi[k] = 1e150;
    for (; n < cap; ++n) {
        const double t = zr * zr - zi * zi;
        zi = __builtin_fma(2.0, zr * zi, ci);
        zr = t + cr;
        if (zr * zr + zi * zi >= 4.0) { *delta = 1e300; return n; }
#pragma unroll
        for (int k = 8; k > 0; --k) { hr[k] = hr[k - 1]; hi[k] = hi[k - 1]; }
        hr[0] = zr;
        hi[0] = zi;
    }
    double d = 1e300;
#pragma unroll
    for (int k = 1; k <= 8; ++k) {
        if (k == 7) continue;
        const double dr = hr[0] - hr[k], di = hi[0] - hi[k], q = dr * dr + di * di;
        d = q < d ? q : d;
    }
    *delta = d;
    return 0;
}

// counters[0..3): H, V units, M; extra[0..3): late M, settled H, a ticket.  With `plan` the workgroup that finishes last turns
// the counts into the XCD shares itself (units_plan: round 4 ran it in a one-thread kernel of its own) and clears the six words for the
// next launch that uses them -- the pre-pass is then ONE kernel instead of fill + classify + plan, three
// dependent launches of a few microseconds each that sit between two tile kernels once a launch no longer ends in a long drain
// (profiles/r05/gaps_*.txt).  The caller zeroes the six words once, when it allocates them.
__global__ __launch_bounds__(1024) void classify_units_kernel(TileArgs p, uint32_t nregions, int32_t probe_steps,
                                                              uint32_t *order, uint32_t *counters, int32_t m_late, double settle_thr,
                                                              uint32_t *extra, uint32_t *plan, XcdShares shares, uint32_t stamps)
{
    __shared__ uint32_t s_last;
    __shared__ uint32_t s_cnt[5][16], s_base[5];
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const bool valid = r < nregions;
    uint32_t cls = 5u;     // 0 H, 1 V, 2 M, 3 M late, 4 H settled, 5 nothing
    uint32_t by = 0, bx = 0;
    if (valid) {
        by = r / p.blocks_x;
        bx = r - by * p.blocks_x;
        uint32_t lc = bx * 8u + 4u, lr = by * 8u + 4u;
        lc = lc < p.ncols ? lc : p.ncols - 1u;
        lr = lr < p.nrows ? lr : p.nrows - 1u;
        const double cr = axis_value(p.re, p.col0 + lc), ci = axis_value(p.im, p.row0 + lr);
        const int32_t cap = p.mrd < probe_steps ? p.mrd : probe_steps;
        double delta = 1e300;
        const int32_t cnt = cap > 1 ? (settle_thr > 0.0 ? probe_settling(cr, ci, cap, &delta) : escape_count<true>(cr, ci, cap)) : 1;
        const bool regular = bx < p.fast_bx_end && by < p.fast_by_end;   // the light path's coordinates are the regular formula
        cls = cnt == 0 ? (delta <= settle_thr ? 4u : 0u) : (cnt <= 3 && regular ? 1u : (m_late > 0 && cnt >= m_late ? 3u : 2u));
    }
    // V blocks -> one unit per aligned group of 8 lanes (= 8 block columns of one row: blocks_x % 8 == 0 and the workgroup's
    // first region is a multiple of 8)
    const unsigned long long vmask = __ballot(cls == 1u);
    const uint32_t seg = (uint32_t)(vmask >> (lane & ~7u)) & 0xffu;
    const bool emit[5] = {cls == 0u, (lane & 7u) == 0u && seg != 0u, cls == 2u, cls == 3u, cls == 4u};
    unsigned long long m[5];
#pragma unroll
    for (uint32_t k = 0; k < 5u; ++k) {
        m[k] = __ballot(emit[k]);
        if (lane == 0) s_cnt[k][wave] = (uint32_t)__popcll(m[k]);
    }
    __syncthreads();
    if (threadIdx.x < 5u) {
        const uint32_t k = threadIdx.x, nw = (blockDim.x + 63u) >> 6;
        uint32_t t = 0;
        for (uint32_t w = 0; w < nw; ++w) {
            const uint32_t c = s_cnt[k][w];
            s_cnt[k][w] = t;
            t += c;
        }
        s_base[k] = t ? atomicAdd(k < 3u ? &counters[k] : &extra[k - 3u], t) : 0u;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    if (emit[0]) order[s_base[0] + s_cnt[0][wave] + (uint32_t)__popcll(m[0] & below)] = (by << 16) | bx;
    if (emit[1]) order[nregions - 1u - (s_base[1] + s_cnt[1][wave] + (uint32_t)__popcll(m[1] & below))] = (by << 16) | ((bx >> 3) << 8) | seg;
    if (emit[2]) order[nregions + 3u + s_base[2] + s_cnt[2][wave] + (uint32_t)__popcll(m[2] & below)] = (by << 16) | bx;
    if (emit[3]) order[2u * nregions + 2u - (s_base[3] + s_cnt[3][wave] + (uint32_t)__popcll(m[3] & below))] = (by << 16) | bx;
    if (emit[4]) order[units_settled_base(nregions) + s_base[4] + s_cnt[4][wave] + (uint32_t)__popcll(m[4] & below)] = (by << 16) | bx;
    if (plan == nullptr) return;
    // the last workgroup to get here: every other one has added its counts (its atomics precede its ticket)
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(&extra[2], 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (s_last != 0u && threadIdx.x == 0) {
        __threadfence();
        const uint32_t n_h = atomicExch(&counters[0], 0u), n_v = atomicExch(&counters[1], 0u), n_m = atomicExch(&counters[2], 0u);
        const uint32_t n_ml = atomicExch(&extra[0], 0u), n_hs = atomicExch(&extra[1], 0u);
        atomicExch(&extra[2], 0u);
        units_plan(n_h + n_ml + n_hs, n_v, n_m, shares.cum, stamps, plan, n_ml, n_hs);
    }
}

// ---- Shares of the eight XCDs ------------------------------------------------------------------------------------------
// The hardware deals workgroup ids to the XCDs in turn (id mod 8 -- profiles/microbench/units_trace.hip read XCC_ID), every XCD
// works through its own ids, and the XCDs of one chip do not run at one speed: equal shares end 2-10 % apart and the launch
// lasts as long as its slowest XCD (profiles/NOTES.md 2b).  Atomics across XCDs are far too slow to rebalance at run time
// (units_pool_ab.txt), so the shares themselves are uneven: XCD x takes h[x] entries of the front list (late M, H, settled H),
// then l[x] of the light list (M entries, then V units).  Workgroup id 8 j + x finds its unit without atomics: entry 8 j + x while j is below the
// smallest share (the even deal, nearly all ids), and behind that the rest of the list in one contiguous piece per XCD
// (entry base[x] + j).  The host sets the H fractions from what earlier launches on the stream reported (mbk_api.hip: the time at
// which every XCD dealt its last ids, kStampTail plain stores per XCD into pinned memory); the counts are known only on the
// device, so the last workgroup of classify turns fractions into shares.  Which XCD computes a block changes when it
// is computed, never what is stored.
constexpr uint32_t kStampTail = 8;                      // ids per XCD, from the end, that leave a time stamp
constexpr uint32_t kStampWords = 8u + 8u * kStampTail;  // per launch: first id of every XCD, then the tails
constexpr uint32_t kPlanWords = 40;  // [0] H entries [1] M entries [2] ids in all (8 per XCD round) [3] min h [4] min l
                                     // [5] 1 = this launch leaves time stamps [6] late M entries: the first [6] of the [0] front
                                     // entries [7] where the settled H entries begin in the front list
                                     // [8..16) h[x]  [16..24) l[x]  [24..32) H base[x]  [32..40) light base[x]

// (host and device: mbk_units_plan / mbk_units_lookup of the C ABI run the same lines for the CPU tests)
// (n_h: the entries of the FRONT list = the n_ml late M entries, the unsettled H entries, the n_hs settled ones; n_m: the other M)
__host__ __device__ inline void units_plan(uint32_t n_h, uint32_t n_v, uint32_t n_m, const uint32_t *cum, uint32_t stamps, uint32_t *plan,
                                           uint32_t n_ml, uint32_t n_hs)
{
    const uint32_t n_l = n_m + n_v, total = n_h + n_l;
    uint32_t h[8], l[8], prev = 0, slots = (total + 7u) >> 3;
    for (uint32_t x = 0; x < 8u; ++x) {
        uint32_t c = x == 7u ? n_h : (uint32_t)(((unsigned long long)n_h * cum[x]) >> 24);
        c = c < prev ? prev : (c > n_h ? n_h : c);
        h[x] = c - prev;
        prev = c;
        slots = slots > h[x] ? slots : h[x];
    }
    // light entries: every XCD is filled up to `slots` ids, the first XCDs first (8 slots >= total: the list fits) -- an
    // XCD with a small H share takes more of them
    uint32_t left = n_l, hmin = 0xffffffffu, lmin = 0xffffffffu;
    for (uint32_t x = 0; x < 8u; ++x) {
        const uint32_t room = slots - h[x];
        l[x] = room < left ? room : left;
        left -= l[x];
        hmin = hmin < h[x] ? hmin : h[x];
        lmin = lmin < l[x] ? lmin : l[x];
    }
    plan[0] = n_h; plan[1] = n_m; plan[2] = slots * 8u; plan[3] = hmin; plan[4] = lmin; plan[5] = stamps; plan[6] = n_ml < n_h ? n_ml : n_h; plan[7] = n_hs < n_h ? n_h - n_hs : 0u;
    uint32_t hb = 8u * hmin, lb = 8u * lmin;   // the contiguous pieces start behind the evenly dealt part
    for (uint32_t x = 0; x < 8u; ++x) {
        plan[8u + x] = h[x];
        plan[16u + x] = l[x];
        plan[24u + x] = hb - hmin;            // entry of id (x, j >= hmin): base + j
        plan[32u + x] = lb - lmin;
        hb += h[x] - hmin;
        lb += l[x] - lmin;
    }
}

// Id u = 8 j + x: false = none (beyond this XCD's shares; so are all later ids of the XCD), else the entry of the H list
// (is_h) or of the light list it takes.
__host__ __device__ __forceinline__ bool units_lookup(uint32_t u, uint32_t hmin, uint32_t lmin, uint32_t h_x, uint32_t l_x,
                                                      uint32_t hbase_x, uint32_t lbase_x, bool &is_h, uint32_t &i)
{
    const uint32_t x = u & 7u, j = u >> 3, k = j - h_x;
    is_h = j < h_x;
    if (!is_h && k >= l_x) return false;
    i = is_h ? (j < hmin ? u : hbase_x + j) : (k < lmin ? 8u * k + x : lbase_x + k);
    return true;
}

// The eleven words of the plan a workgroup of XCD x needs, with scalar loads issued together (scalar_load_u32's comment)
__device__ __forceinline__ void load_shares(const uint32_t *plan, uint32_t x, uint32_t &n_m, uint32_t &total, uint32_t &hmin,
                                            uint32_t &lmin, uint32_t &stamps, uint32_t &n_ml, uint32_t &hs0, uint32_t &h_x, uint32_t &l_x,
                                            uint32_t &hbase_x, uint32_t &lbase_x)
{
    const uint32_t *pl = plan, *px = plan + x;      // (kernel argument + workgroup id: scalar registers as they stand)
    asm volatile("s_load_dword %0, %11, 0x4\n\t"
                 "s_load_dword %1, %11, 0x8\n\t"
                 "s_load_dword %2, %11, 0xc\n\t"
                 "s_load_dword %3, %11, 0x10\n\t"
                 "s_load_dword %4, %11, 0x14\n\t"
                 "s_load_dword %5, %11, 0x18\n\t"
                 "s_load_dword %6, %11, 0x1c\n\t"
                 "s_load_dword %7, %12, 0x20\n\t"
                 "s_load_dword %8, %12, 0x40\n\t"
                 "s_load_dword %9, %12, 0x60\n\t"
                 "s_load_dword %10, %12, 0x80\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(n_m), "=&s"(total), "=&s"(hmin), "=&s"(lmin), "=&s"(stamps), "=&s"(n_ml), "=&s"(hs0), "=&s"(h_x), "=&s"(l_x),
                   "=&s"(hbase_x), "=&s"(lbase_x)
                 : "s"(pl), "s"(px)
                 : "memory");
}

// One time stamp for the host's shares (100 MHz wall clock, launch number on top) into word `slot` of the launch's record.
// Reached by 72 workgroups of a launch: the two arguments are fetched HERE, from the kernel argument segment, and the lane
// number is derived here, so that none of it occupies a register anywhere else.
__device__ __forceinline__ void leave_stamp(uint32_t slot)
{
    unsigned long long base;
    uint32_t tag;
    asm volatile("s_load_dwordx2 %0, %2, %3\n\t"
                 "s_load_dword %1, %2, %4\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(base), "=&s"(tag)
                 : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(offsetof(TileArgs, stamps)), "n"(offsetof(TileArgs, stamp_tag))
                 : "memory");
    if (__lane_id() == 0)
        reinterpret_cast<unsigned long long *>(base)[slot] = (wall_clock64() & 0xffffffffffffull) | ((unsigned long long)tag << 48);
}

// Workgroup b: ids b, b + G, ... (G = p.unit_stride: the grid size, a multiple of 8, passed as an argument -- gridDim.x
// lives in the dispatch packet in host memory and would be re-read on every trip); id 8 j + x is the j-th of XCD x.  kGroup: 16 (fp64) / 8 (fp32), as in tile_asm_kernel.
// kStats (bytes-only instantiation): the kernel adds the tile's pixel-iterations and never-escaped count to args.stats itself --
// per lane in registers over the wave's units, one reduction and two atomics per wave -- so that a DataChunk whose caller
// wants bytes only writes no int32 counts and the statistics pass reads the 16 MiB of bytes only (what the finish-in-place
// light pass of kernel "scan" does for all-exterior tiles, here for the tiles that hold part of the set).
template <typename T, int kGroup, bool kCycle, bool kCounts, bool kBytes, bool kStats = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void tile_units_kernel(TileArgs args, uint32_t qtab)
{
    static_assert(!kStats || (!kCounts && kBytes), "fused statistics exist for the bytes-only instantiation");
    // Statistics of the blocks computed by block_pixel, per lane, in ONE 64-bit register (this instantiation has no vector
    // register to spare at 8 waves per SIMD): pixel-iterations in bits 0..46, never-escaped pixels above.  A wave computes at
    // most 8 blocks per trip and makes at most 2^13 trips (grid.x < 2^27 regions, the stride at least 2^14 ids): fewer than
    // 2^16 blocks, under 2^31 iterations each.
    unsigned long long heavy = 0;
    uint32_t acc = 0;                       // counts of the blocks the light path finished (<= 4 each)
    const uint32_t never_cap = args.mrd > 1 ? (uint32_t)args.mrd - 1u : 0u;
    // The loop keeps the launch's arguments alive across a whole block, and the escape loops need their share of the 96
    // scalar registers that 8 waves per SIMD leave a wave.  What this kernel never uses is pinned to the value the host
    // guarantees (launch_blocks: no smooth output, no 64-bit quantiser, non-zero steps, no fused statistics), so that the
    // code -- and the registers -- for it disappear.
    TileArgs p = args;
    p.smooth = nullptr;
    p.stats = nullptr;
    p.quant_wide = 0u;
    p.re.step_is_zero = p.im.step_is_zero = 0u;
    if (!kCounts) p.counts = nullptr;
    if (!kBytes) p.bytes = nullptr;
    const uint32_t lane = threadIdx.x;
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const uint32_t n = p.ngrid;
    const uint32_t x = blockIdx.x & 7u;                      // this workgroup's XCD (the stride is a multiple of 8)
    const uint32_t oscale = kCounts ? 4u : 1u;
    for (uint32_t u = blockIdx.x;; u += p.unit_stride) {
        // (the shares are read anew on every trip -- a second trip is rare -- so that nothing of them stays in scalar
        // registers across a block)
        uint32_t n_m, total, hmin, lmin, stamps, n_ml, hs0, h_x, l_x, hbase_x, lbase_x;
        load_shares(args.plan, x, n_m, total, hmin, lmin, stamps, n_ml, hs0, h_x, l_x, hbase_x, lbase_x);
        if (stamps && u < p.unit_stride) {
            // first trip: the first workgroup of every XCD and its last kStampTail tell the host when they started
            const uint32_t j0 = u >> 3, slots = total >> 3;
            if (j0 == 0u) leave_stamp(x);
            if (j0 < slots && j0 + kStampTail >= slots) leave_stamp(8u + x * kStampTail + (j0 + kStampTail - slots));
        }
        if (u >= total) break;
        bool is_h;
        uint32_t i;    // index into the H list / the light list (M entries, then V units)
        if (!units_lookup(u, hmin, lmin, h_x, l_x, hbase_x, lbase_x, is_h, i)) break;
        if (is_h || i < n_m) {
            // (front entry i: the late M entries from the back of the M region, the H list, the settled H entries from their region)
            const bool late = is_h && i < n_ml;
            const uint32_t e = scalar_load_u32(p.order, is_h ? (late ? 2u * n + 2u - i : (i < hs0 ? i - n_ml : (uint32_t)units_settled_base(n) + (i - hs0)))
                                                              : n + 3u + i);
            const uint32_t by = e >> 16, bx = e & 0xffffu;
            const int32_t c = block_pixel<T, true, kGroup, kCycle>(p, bx * 8u, by * 8u, lx, ly, kGroup >= 16 && is_h && !late,
                                                                   bx < p.fast_bx_end && by < p.fast_by_end);
            if (kStats && c >= 0) heavy += c > 0 ? (unsigned long long)(uint32_t)c : (1ull << 47) + never_cap;
        } else {
            const uint32_t v = scalar_load_u32(p.order, n - 1u - (i - n_m));
            const uint32_t by = v >> 16, bx0 = ((v >> 8) & 0xffu) << 3, mask = v & 0xffu;
            // the unit's imaginary coordinate (regular formula: classify files under V only blocks inside the fast region)
            const T ci = (T)((double)(p.row0 + by * 8u + ly) * p.im.step + p.im.start);
            const T b0 = ci * ci;
            const size_t elem0 = (size_t)(by * 8u + p.out_row0) * p.out_pitch + bx0 * 8u + p.out_col0;
            int32_t *cb = reinterpret_cast<int32_t *>(uniform_u64(reinterpret_cast<unsigned long long>(kCounts ? p.counts + elem0 : nullptr)));
            uint8_t *bb = reinterpret_cast<uint8_t *>(uniform_u64(reinterpret_cast<unsigned long long>(kBytes ? p.bytes + elem0 : nullptr)));
            const uint32_t col = p.col0 + bx0 * 8u + lx, off = (ly * p.out_pitch + lx) * oscale;
            uint32_t k = 0;
            int32_t cnt;
            while (escape_light_row<kCounts, kBytes, kStats>(ci, b0, col, p.re.step, p.re.start, cnt, cb, bb, off, 8u * oscale, qtab, mask, k, &acc) != 0u) {
                // block k of the unit outlives the light path (the probe saw only its centre pixel): the whole block, exactly
                const int32_t c = block_pixel<T, true, kGroup, kCycle>(p, (bx0 + k) * 8u, by * 8u, lx, ly, false, true);
                if (kStats && c >= 0) heavy += c > 0 ? (unsigned long long)(uint32_t)c : (1ull << 47) + never_cap;
                if (++k >= 8u) break;
            }
        }
    }
    if (kStats) {
        // (acc cannot wrap: <= 4 per block, and a wave handles far fewer than 2^29 blocks)
        const unsigned long long iters = wave_sum_u64((heavy & ((1ull << 47) - 1ull)) + acc), never = wave_sum_u64(heavy >> 47);
        ReduceOut *out = &args.stats[blockIdx.x % kReduceSlots].r;
        if (lane == 0) {
            if (iters) atomicAdd(&out->pixel_iterations, iters);
            if (never) atomicAdd(&out->never_pixels, never);
        }
    }
}

}  // namespace mbk

// mbk_refill.h -- kernel "refill": persistent wavefronts with lane refill, for gfx950.
//
// Why (measured on MI355X, BASELINE cfg2): with one workgroup per 32x8 pixel block
// the hand-scheduled loop reaches 4.4 T pixel-iter/s on a uniform all-in-set tile but only ~2.9 T on
// cfg2: 83 % of the 65 536 workgroups are trivial (dispatch-bound), the heavy ones cluster in space
// so the in-order dispatcher leaves SIMDs under-occupied (average 3.9 of 8 waves), lanes whose pixel
// escaped idle until the slowest lane of their wave is done, and the last heavy workgroups drain on
// an almost empty chip.  Here instead:
//   * the grid is sized to fill the machine once (8 waves/SIMD) and every wave is a worker that
//     lives until the tile is done -- no dispatch cost per block, no clustered tail;
//   * pixels are handed out in 8x8 blocks from 64 small work queues in HBM (one atomicAdd per 64
//     pixels, 64 addresses so no single counter is hot; a 64-bit "non-empty" mask lets a worker whose
//     home queue ran dry find the remaining work in O(1)) -- this is the per-GPU dynamic work queue of
//     the north star pushed down to wavefront granularity;
//   * a lane whose pixel escaped is refilled with the next pixel of the wave's current block while
//     the other lanes keep iterating: per-lane state is just (c, z, |z|^2 parts, start clock);
//   * the iteration "clock" n is wave-uniform (SGPR); a lane's escape index is clock - start.
// The hot loop is the same hand-scheduled stream as kernel "asm" (7 fp64 VALU + v_cmp + 1 branch per
// step, 4 SALU per 4 steps); everything rare (retire, refill, queue pops, the mrd deadline) is plain
// HIP C++ between two entries of the loop.
//
// Bit-exactness: identical arithmetic per pixel as the other kernels; only the ORDER in which pixels
// are processed differs, and pixels are independent.  A lane may run up to 3 steps past mrd-1
// (the deadline is checked at 4-step boundaries); an "escape" recorded past mrd-1 is reported as 0.
#pragma once

#include "mbk_kernels.h"

namespace mbk {

constexpr int kNumQueues = 64;
constexpr uint32_t kNoBlock = 0xffffffffu;

struct WorkQueues {  // 64-byte slots: each counter on its own cache line
    struct Slot {
        unsigned int next;
        unsigned int pad[15];
    } q[kNumQueues];
};

// queue t owns blocks [queue_lo(t), queue_lo(t+1)) -- pure arithmetic, so `end` is never loaded
__device__ __forceinline__ uint32_t queue_lo(uint32_t nblocks, uint32_t t)
{
    return (uint32_t)(((uint64_t)nblocks * t) / kNumQueues);
}

// One block of 64 threads resets the 64 cursors (runs on the launch stream before the tile kernel).
__global__ __launch_bounds__(64) void init_queues_kernel(WorkQueues *w, uint32_t nblocks)
{
    w->q[threadIdx.x].next = queue_lo(nblocks, threadIdx.x);
}

// Per-wave view of the queues: the queue it is currently draining and that queue's end.
struct Popper {
    uint32_t cq;      // current queue (wave-uniform)
    uint32_t cq_end;  // its end
};

// Pop `want` consecutive 8x8 blocks for this wave: returns the first block id (or kNoBlock when the
// whole tile has been handed out) and the number granted in *got.  All 64 lanes call it.
// Fast path: ONE atomicAdd on the wave's current queue (64 addresses, ~128 customers each).
// When that queue is dry: every lane loads one queue cursor (one vector load, 64 lines), the ballot
// of "cursor < end" is a fresh non-empty mask, and the wave moves to the nearest non-empty queue.
__device__ __forceinline__ uint32_t pop_blocks(WorkQueues *w, Popper &pp, uint32_t nblocks,
                                               uint32_t home, uint32_t lane, uint32_t want,
                                               uint32_t *got)
{
    for (;;) {
        uint32_t idx = 0;
        if (lane == 0) idx = atomicAdd(&w->q[pp.cq].next, want);
        idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        if (idx < pp.cq_end) {
            const uint32_t left = pp.cq_end - idx;
            *got = left < want ? left : want;
            return idx;
        }
        // current queue is dry: scan all cursors at once
        const uint32_t nx = __hip_atomic_load(&w->q[lane].next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long m = __ballot(nx < queue_lo(nblocks, lane + 1u));
        if (m == 0) {
            *got = 0;
            return kNoBlock;
        }
        const unsigned long long rot = (m >> home) | (m << ((64u - home) & 63u));
        pp.cq = (home + (uint32_t)__ffsll((long long)rot) - 1u) & 63u;
        pp.cq_end = queue_lo(nblocks, pp.cq + 1u);
    }
}

__device__ __forceinline__ uint32_t uniform_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v)
{
    const uint32_t lo = uniform_u32((uint32_t)v), hi = uniform_u32((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// ---- the hot loop: runs the lanes in LIVE until an event needs the slow path ------------------
// Leaves when (a) no lane is left, (b) at most LIVEMIN lanes are left (enough free lanes to make a
// refill worthwhile), (c) the clock reaches ALARM at a 4-step boundary (ALARM = mrd deadline bound,
// pulled in to "first unrefilled escape + PATIENCE").  Escaped lanes get CNT = clock - START.
#define MBK_RF_HEAD_FMA                                    \
    "v_add_f64 %[t], %[a], -%[b]\n"                        \
    "v_mul_f64 %[p], %[zr], %[zi]\n"                       \
    "v_add_f64 %[zr], %[t], %[cr]\n"                       \
    "v_fma_f64 %[zi], %[p], 2.0, %[ci]\n"
#define MBK_RF_HEAD_SAFE                                   \
    "v_add_f64 %[t], %[a], -%[b]\n"                        \
    "v_add_f64 %[p], %[zr], %[zr]\n"                       \
    "v_mul_f64 %[p], %[p], %[zi]\n"                        \
    "v_add_f64 %[zr], %[t], %[cr]\n"                       \
    "v_add_f64 %[zi], %[p], %[ci]\n"
#define MBK_RF_STEP(ID)                                    \
    "v_mul_f64 %[a], %[zr], %[zr]\n"                       \
    "v_mul_f64 %[b], %[zi], %[zi]\n"                       \
    "v_add_f64 %[m], %[a], %[b]\n"                         \
    "v_cmp_le_f64 vcc, 4.0, %[m]\n"                        \
    "s_cbranch_vccnz .Lresc" ID "_%=\n"                    \
    ".Lrcont" ID "_%=:\n"
#define MBK_RF_ESCAPE(ID, INC)                             \
    ".Lresc" ID "_%=:\n"                                   \
    "s_add_u32 %[k], %[n], " INC "\n"                      \
    "s_and_saveexec_b64 %[tmp], vcc\n"                     \
    "v_sub_u32 %[cnt], %[k], %[start]\n"                   \
    "s_andn2_b64 exec, %[tmp], vcc\n"                      \
    "s_cbranch_scc0 .Lrmid_%=\n"                           \
    "s_bcnt1_i32_b64 %[k2], exec\n"                        \
    "s_cmp_le_u32 %[k2], %[livemin]\n"                     \
    "s_cbranch_scc1 .Lrmid_%=\n"                           \
    "s_add_u32 %[k2], %[k], %[patience]\n"                 \
    "s_sub_u32 %[k3], %[k2], %[alarm]\n"                   \
    "s_cmp_lt_i32 %[k3], 0\n"                              \
    "s_cselect_b32 %[alarm], %[k2], %[alarm]\n"            \
    "s_branch .Lrcont" ID "_%=\n"
#define MBK_RF_LOOP(HEAD)                                  \
    "s_mov_b64 %[save], exec\n"                            \
    "s_mov_b64 exec, %[live]\n"                            \
    ".Lrmain_%=:\n"                                        \
    HEAD MBK_RF_STEP("1") HEAD MBK_RF_STEP("2")            \
    HEAD MBK_RF_STEP("3") HEAD MBK_RF_STEP("4")            \
    "s_add_u32 %[n], %[n], 4\n"                            \
    "s_sub_u32 %[k3], %[n], %[alarm]\n"                    \
    "s_cmp_lt_i32 %[k3], 0\n"                              \
    "s_cbranch_scc1 .Lrmain_%=\n"                          \
    "s_branch .Lrout_%=\n"                                 \
    MBK_RF_ESCAPE("1", "1") MBK_RF_ESCAPE("2", "2")        \
    MBK_RF_ESCAPE("3", "3") MBK_RF_ESCAPE("4", "4")        \
    ".Lrmid_%=:\n"                                         \
    "s_mov_b32 %[n], %[k]\n"                               \
    ".Lrout_%=:\n"                                         \
    "s_mov_b64 %[live], exec\n"                            \
    "s_mov_b64 exec, %[save]\n"

// ---- grouped hot loop for the persistent kernel -------------------------------------------------
// Same idea as escape_steps_group (mbk_loops.inc): 8 unchecked steps on a scratch register set, one
// NaN-inclusive test per group, exact replay for the lanes that tripped it (cnt = clock - start),
// two register sets A/B alternating so a group's start state survives until its test.  One trip =
// 16 steps.  The slow-path exits (too few live lanes / alarm) happen after a replay or at the end of
// a trip; if the wave leaves after the first group the live state is copied back from set B to set A,
// which is where the C++ side keeps it.
#define MBK_RFG_STEP(ZRS, ZIS, AS, BS, ZRD, ZID, AD, BD)   \
    "v_add_f64 %[t], " AS ", -" BS "\n"                    \
    "v_mul_f64 %[p], " ZRS ", " ZIS "\n"                   \
    "v_add_f64 " ZRD ", %[t], %[cr]\n"                     \
    "v_fma_f64 " ZID ", %[p], 2.0, %[ci]\n"                \
    "v_mul_f64 " AD ", " ZRD ", " ZRD "\n"                 \
    "v_mul_f64 " BD ", " ZID ", " ZID "\n"
#define MBK_RFG_A2T MBK_RFG_STEP("%[zr]", "%[zi]", "%[a]", "%[b]", "%[zrt]", "%[zit]", "%[at]", "%[bt]")
#define MBK_RFG_B2T MBK_RFG_STEP("%[zr2]", "%[zi2]", "%[a2]", "%[b2]", "%[zrt]", "%[zit]", "%[at]", "%[bt]")
#define MBK_RFG_T2T MBK_RFG_STEP("%[zrt]", "%[zit]", "%[at]", "%[bt]", "%[zrt]", "%[zit]", "%[at]", "%[bt]")
#define MBK_RFG_T2A MBK_RFG_STEP("%[zrt]", "%[zit]", "%[at]", "%[bt]", "%[zr]", "%[zi]", "%[a]", "%[b]")
#define MBK_RFG_T2B MBK_RFG_STEP("%[zrt]", "%[zit]", "%[at]", "%[bt]", "%[zr2]", "%[zi2]", "%[a2]", "%[b2]")
#define MBK_RFG_GROUP8(FIRST, LAST, AD, BD, ID)            \
    FIRST MBK_RFG_T2T MBK_RFG_T2T MBK_RFG_T2T MBK_RFG_T2T MBK_RFG_T2T MBK_RFG_T2T LAST \
    "v_add_f64 %[m], " AD ", " BD "\n"                     \
    "v_cmp_ngt_f64 vcc, 4.0, %[m]\n"                       \
    "s_cbranch_vccnz .Lqrep" ID "_%=\n"                    \
    ".Lqcont" ID "_%=:\n"
#define MBK_RFG_REPLAY_STEP(STEP, J)                       \
    STEP                                                   \
    "v_add_f64 %[m], %[at], %[bt]\n"                       \
    "v_cmp_le_f64 vcc, 4.0, %[m]\n"                        \
    "s_add_u32 %[k], %[n], " J "\n"                        \
    "s_or_b64 %[esc], %[esc], vcc\n"                       \
    "s_and_saveexec_b64 %[tmp2], vcc\n"                    \
    "v_sub_u32 %[cnt], %[k], %[start]\n"                   \
    "s_andn2_b64 exec, %[tmp2], vcc\n"
#define MBK_RFG_REPLAY8(FIRST, ID, J1, J2, J3, J4, J5, J6, J7, J8, NADV, COPYBACK) \
    ".Lqrep" ID "_%=:\n"                                   \
    "s_and_saveexec_b64 %[tmp], vcc\n"                     \
    "s_mov_b64 %[esc], 0\n"                                \
    MBK_RFG_REPLAY_STEP(FIRST, J1) MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J2) \
    MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J3) MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J4) \
    MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J5) MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J6) \
    MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J7) MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J8) \
    "s_andn2_b64 exec, %[tmp], %[esc]\n"                   \
    "s_cbranch_scc0 .Lqexit" ID "_%=\n"                    \
    "s_bcnt1_i32_b64 %[k2], exec\n"                        \
    "s_cmp_le_u32 %[k2], %[livemin]\n"                     \
    "s_cbranch_scc1 .Lqexit" ID "_%=\n"                    \
    "s_add_u32 %[k2], %[n], " NADV "\n"                    \
    "s_add_u32 %[k2], %[k2], %[patience]\n"                \
    "s_sub_u32 %[k3], %[k2], %[alarm]\n"                   \
    "s_cmp_lt_i32 %[k3], 0\n"                              \
    "s_cselect_b32 %[alarm], %[k2], %[alarm]\n"            \
    "s_branch .Lqcont" ID "_%=\n"                          \
    ".Lqexit" ID "_%=:\n"                                  \
    COPYBACK                                               \
    "s_add_u32 %[n], %[n], " NADV "\n"                     \
    "s_branch .Lqout_%=\n"
#define MBK_RFG_COPY_B2A                                   \
    "v_mov_b64 %[zr], %[zr2]\n"                            \
    "v_mov_b64 %[zi], %[zi2]\n"                            \
    "v_mov_b64 %[a], %[a2]\n"                              \
    "v_mov_b64 %[b], %[b2]\n"
#define MBK_RFG_LOOP                                       \
    "s_mov_b64 %[save], exec\n"                            \
    "s_mov_b64 exec, %[live]\n"                            \
    ".Lqmain_%=:\n"                                        \
    MBK_RFG_GROUP8(MBK_RFG_A2T, MBK_RFG_T2B, "%[a2]", "%[b2]", "1") \
    MBK_RFG_GROUP8(MBK_RFG_B2T, MBK_RFG_T2A, "%[a]", "%[b]", "2")   \
    "s_add_u32 %[n], %[n], 16\n"                           \
    "s_sub_u32 %[k3], %[n], %[alarm]\n"                    \
    "s_cmp_lt_i32 %[k3], 0\n"                              \
    "s_cbranch_scc1 .Lqmain_%=\n"                          \
    "s_branch .Lqout_%=\n"                                 \
    MBK_RFG_REPLAY8(MBK_RFG_A2T, "1", "1", "2", "3", "4", "5", "6", "7", "8", "8", MBK_RFG_COPY_B2A)   \
    MBK_RFG_REPLAY8(MBK_RFG_B2T, "2", "9", "10", "11", "12", "13", "14", "15", "16", "16", "")         \
    ".Lqout_%=:\n"                                         \
    "s_mov_b64 %[live], exec\n"                            \
    "s_mov_b64 exec, %[save]\n"

// per-lane predicate from a wave-uniform mask without 64-bit VALU shifts
__device__ __forceinline__ bool lane_in(unsigned long long uniform_mask)
{
    return __builtin_amdgcn_inverse_ballot_w64(uniform_mask);
}
// rank of this lane among the set bits below it
__device__ __forceinline__ uint32_t rank_in(unsigned long long uniform_mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(uniform_mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)uniform_mask, 0u));
}

template <bool kFmaDouble, bool kGrouped = false>
__global__ __launch_bounds__(256) void tile_refill_kernel(TileArgs p, WorkQueues *wq)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t home = (blockIdx.x * 4u + (threadIdx.x >> 6)) & 63u;
    const uint32_t total = (uint32_t)p.mrd - 1u;  // host guarantees mrd >= 2
    const uint32_t bxn = (p.ncols + 7u) / 8u;     // 8x8 blocks per block-row
    constexpr uint32_t kFar = 0x40000000u;         // all relative clock offsets stay below 2^30
    const bool want_bytes = p.bytes != nullptr;

    double cr = 0.0, ci = 0.0, zr = 0.0, zi = 0.0, a = 0.0, b = 0.0;
    uint32_t start = 0, cnt = 0, opix = 0;
    bool risky = false;           // | |c|^2 - 4 | < 1e-9: no grouped test while such a pixel is live
    unsigned long long live = 0;  // wave-uniform: lanes with a pixel in flight
    unsigned long long live_in = 0;
    uint32_t n = 0;               // wave-uniform clock: steps executed by this wave so far
    uint32_t bound = 0;           // lower bound on the earliest clock at which a live lane hits mrd-1
    // current block (all wave-uniform): pixel origin, next pixel in it, blocks still owned, and whether
    // the cheap refill path applies (block wholly inside the window, away from both axis end points)
    uint32_t blk = 0, blk_col = 0, blk_row = 0, blk_pos = 64, blk_left = 0;
    bool blk_simple = false;
    bool more = true;
    const uint32_t nblocks = bxn * ((p.nrows + 7u) / 8u);
    const bool axes_simple = !p.re.step_is_zero && !p.im.step_is_zero;
    Popper pp;
    pp.cq = home;
    pp.cq_end = queue_lo(nblocks, home + 1u);

    for (;;) {
        // ---------------- retire lanes that escaped during the last run ------------------------------
        const unsigned long long finished = live_in & ~live;
        if (finished != 0 && lane_in(finished)) {
            const int32_t count = cnt <= total ? (int32_t)cnt : 0;  // an escape past mrd-1 is "never"
            if (p.counts) p.counts[opix] = count;
            if (want_bytes) p.bytes[opix] = quantise(count, p.mrd, p.quant_wide);
        }
        live_in = live;
        // ---------------- mrd deadline: lanes that ran mrd-1 steps without escaping -> 0 -------------
        if ((int32_t)(n - bound) >= 0) {
            const bool is_live = lane_in(live);
            const uint32_t age = n - start;
            const bool expired = is_live && age >= total;
            if (expired) {
                if (p.counts) p.counts[opix] = 0;
                if (want_bytes) p.bytes[opix] = 0;
            }
            live &= ~__ballot(expired);
            uint32_t rem = (is_live && !expired) ? total - age : kFar;
            rem = wave_min_u32(rem < kFar ? rem : kFar);
            bound = n + uniform_u32(rem);
        }
        // ---------------- refill ----------------------------------------------------------------------
        const bool was_empty = (live == 0);
        unsigned long long free_lanes = ~live;
        while (more && free_lanes != 0) {
            if (blk_pos >= 64u) {  // need the next block
                if (blk_left > 0) {
                    ++blk;
                    --blk_left;
                } else {
                    uint32_t got;
                    blk = pop_blocks(wq, pp, nblocks, home, lane, p.rf_batch, &got);
                    if (blk == kNoBlock) {
                        more = false;
                        break;
                    }
                    blk_left = got - 1u;
                }
                const uint32_t by = blk / bxn, bx = blk - by * bxn;  // scalar ALU (uniform)
                blk_col = bx * 8u;
                blk_row = by * 8u;
                blk_pos = 0;
                blk_simple = axes_simple && blk_col + 8u <= p.ncols && blk_row + 8u <= p.nrows &&
                             p.col0 + blk_col + 8u < p.re.n && p.row0 + blk_row + 8u < p.im.n;
            }
            const uint32_t navail = 64u - blk_pos;
            const uint32_t nfree = (uint32_t)__popcll(free_lanes);
            if (blk_simple && nfree <= navail) {
                // fast path (the common case): every free lane takes a pixel of this interior block;
                // coordinates by the two-rounding linspace formula, no end-point / edge handling needed
                if (lane_in(free_lanes)) {
                    const uint32_t pidx = blk_pos + rank_in(free_lanes);
                    const uint32_t lc = blk_col + (pidx & 7u), lr = blk_row + (pidx >> 3);
                    cr = (double)(p.col0 + lc) * p.re.step + p.re.start;
                    ci = (double)(p.row0 + lr) * p.im.step + p.im.start;
                    zr = cr;
                    zi = ci;
                    a = zr * zr;
                    b = zi * zi;
                    start = n;
                    cnt = 0;
                    opix = lr * p.ncols + lc;
                    const double c2 = a + b;
                    risky = c2 > 4.0 - 1e-9 && c2 < 4.0 + 1e-9;
                }
                live = ~0ull;
                blk_pos += nfree;
                break;
            }
            // general path: block edges, axis end points, or the block runs out mid-refill
            const bool is_free = lane_in(free_lanes);
            const uint32_t rank = rank_in(free_lanes);
            const bool take = is_free && rank < navail;
            bool valid = false;
            if (take) {
                const uint32_t pidx = blk_pos + rank;
                const uint32_t lc = blk_col + (pidx & 7u), lr = blk_row + (pidx >> 3);
                if (lc < p.ncols && lr < p.nrows) {
                    cr = axis_value(p.re, p.col0 + lc);
                    ci = axis_value(p.im, p.row0 + lr);
                    zr = cr;
                    zi = ci;
                    a = zr * zr;
                    b = zi * zi;
                    start = n;
                    cnt = 0;
                    opix = lr * p.ncols + lc;
                    const double c2 = a + b;
                    risky = c2 > 4.0 - 1e-9 && c2 < 4.0 + 1e-9;
                    valid = true;
                }
            }
            const unsigned long long taken = __ballot(take);
            live |= __ballot(valid);
            blk_pos += (uint32_t)__popcll(taken);
            free_lanes = ~live;  // lanes whose pixel fell outside the window try again
        }
        if (live == 0) {
            if (!more) break;
            continue;  // every pixel taken this round was outside the window (ragged edge block)
        }
        if (was_empty) bound = n + (total < kFar ? total : kFar);

        // ---------------- run ---------------------------------------------------------------------
        live_in = live;
        {
            double t, pr, m;
            uint32_t k, k2, k3;
            unsigned long long save, tmp;
            // all of these are wave-uniform by construction; say so to the register allocator
            uint32_t alarm = uniform_u32(bound);
            const uint32_t livemin = uniform_u32(more ? p.rf_livemin : 0u);     // refill once enough lanes are free
            const uint32_t patience = uniform_u32(more ? p.rf_patience : kFar);   // ... or this long after an escape
            n = uniform_u32(n);
            live = uniform_u64(live);
            const bool any_risky = __ballot(risky && lane_in(live)) != 0;
            if (kGrouped && kFmaDouble && !any_risky) {
                double zr2, zi2, a2, b2, zrt, zit, at, bt;
                unsigned long long tmp2, esc;
                asm volatile(MBK_RFG_LOOP
                             : [zr] "+&v"(zr), [zi] "+&v"(zi), [a] "+&v"(a), [b] "+&v"(b),
                               [cnt] "+&v"(cnt), [zr2] "=&v"(zr2), [zi2] "=&v"(zi2), [a2] "=&v"(a2),
                               [b2] "=&v"(b2), [zrt] "=&v"(zrt), [zit] "=&v"(zit), [at] "=&v"(at),
                               [bt] "=&v"(bt), [t] "=&v"(t), [p] "=&v"(pr), [m] "=&v"(m),
                               [n] "+&s"(n), [live] "+&s"(live), [alarm] "+&s"(alarm),
                               [k] "=&s"(k), [k2] "=&s"(k2), [k3] "=&s"(k3), [save] "=&s"(save),
                               [tmp] "=&s"(tmp), [tmp2] "=&s"(tmp2), [esc] "=&s"(esc)
                             : [cr] "v"(cr), [ci] "v"(ci), [start] "v"(start), [livemin] "s"(livemin),
                               [patience] "s"(patience)
                             : "vcc", "scc");
            } else if (kFmaDouble) {
                asm volatile(MBK_RF_LOOP(MBK_RF_HEAD_FMA)
                             : [zr] "+&v"(zr), [zi] "+&v"(zi), [a] "+&v"(a), [b] "+&v"(b),
                               [cnt] "+&v"(cnt), [t] "=&v"(t), [p] "=&v"(pr), [m] "=&v"(m),
                               [n] "+&s"(n), [live] "+&s"(live), [alarm] "+&s"(alarm),
                               [k] "=&s"(k), [k2] "=&s"(k2), [k3] "=&s"(k3), [save] "=&s"(save),
                               [tmp] "=&s"(tmp)
                             : [cr] "v"(cr), [ci] "v"(ci), [start] "v"(start), [livemin] "s"(livemin),
                               [patience] "s"(patience)
                             : "vcc", "scc");
            } else {
                asm volatile(MBK_RF_LOOP(MBK_RF_HEAD_SAFE)
                             : [zr] "+&v"(zr), [zi] "+&v"(zi), [a] "+&v"(a), [b] "+&v"(b),
                               [cnt] "+&v"(cnt), [t] "=&v"(t), [p] "=&v"(pr), [m] "=&v"(m),
                               [n] "+&s"(n), [live] "+&s"(live), [alarm] "+&s"(alarm),
                               [k] "=&s"(k), [k2] "=&s"(k2), [k3] "=&s"(k3), [save] "=&s"(save),
                               [tmp] "=&s"(tmp)
                             : [cr] "v"(cr), [ci] "v"(ci), [start] "v"(start), [livemin] "s"(livemin),
                               [patience] "s"(patience)
                             : "vcc", "scc");
            }
        }
    }
}

}  // namespace mbk

// mbk_persist.h -- kernel "refill" (v4): persistent wavefronts with lane refill for deep zooms.
//
// Why: on a deep zoom (BASELINE cfg3) the pixels of an 8x8 block escape at very different steps; with one
// pixel per lane for the life of the wave only 77 % of the lanes are active on average
// (SQ_THREAD_CYCLES_VALU / (64 SQ_INSTS_VALU)), and a CPU simulation on the real counts says that
// refilling freed lanes brings that to ~0.92 PROVIDED a refill event costs less than ~12 loop steps.
// So this kernel is deliberately narrow -- everything that would cost registers or instructions in the
// slow path is excluded by the host (mbk_api.hip: launch_refill) and served by the `group` kernel:
//   * only INTERIOR 8x8 blocks (wholly inside the window, away from the end point of either axis, so
//     the coordinate is always fl(fl(k*step)+start));  the right / bottom edge strips go to `group`;
//   * only views that cannot touch the |c| = 2 ring (no per-lane "risky" state, grouped test always);
//   * only the fma(2, zr*zi, ci) form (no tiny imaginary parts), counts only (bytes by a post-pass).
// Structure: the grid fills the chip once; 64 cursors in HBM hand out blocks (one atomicAdd per pop);
// the hot loop is the grouped stream of mbk_loops.inc (8 unchecked steps, NaN-inclusive test, exact
// replay) with a wave-uniform clock: a lane's count is clock - start.  Slow path = retire (one store),
// refill (rank among free lanes -> pixel of the current block -> 2 coordinates -> z = c), deadline.
// Bit-exactness: same arithmetic per pixel as every other kernel; only the processing ORDER differs.
#pragma once

#include "mbk_refill.h"  // WorkQueues, pop_blocks, the MBK_RFG_* loop macros, helpers

namespace mbk {

struct PersistArgs {
    double re_start, re_step, im_start, im_step;  // x[k] = fl(fl(k*step)+start), k = col0 + column
    uint32_t col0, row0;                           // window origin inside the view
    uint32_t pitch;                                // output elements per row (= window columns)
    uint32_t bxn, nblocks;                         // interior blocks per block-row, total
    uint32_t total;                                // mrd - 1 (>= 1)
    uint32_t livemin, patience, batch;             // refill policy / blocks per pop
    int32_t *counts;
};

__global__ __launch_bounds__(256) void tile_persist_kernel(PersistArgs p, WorkQueues *wq)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t home = (blockIdx.x * 4u + (threadIdx.x >> 6)) & 63u;
    constexpr uint32_t kFar = 0x40000000u;  // all relative clock offsets stay below 2^30

    double cr = 0.0, ci = 0.0, zr = 0.0, zi = 0.0, a = 0.0, b = 0.0;
    uint32_t start = 0, cnt = 0, opix = 0;
    unsigned long long live = 0, live_in = 0;  // wave-uniform lane masks
    uint32_t n = 0;                            // wave-uniform clock
    uint32_t bound = 0;                        // lower bound on the first clock a live lane reaches `total`
    uint32_t blk = 0, blk_col = 0, blk_row = 0, blk_pos = 64, blk_left = 0;
    bool more = true;
    Popper pp;
    pp.cq = home;
    pp.cq_end = queue_lo(p.nblocks, home + 1u);

    for (;;) {
        // ---- retire: lanes that escaped during the last run (cnt > total: escaped past mrd-1 -> 0)
        const unsigned long long finished = live_in & ~live;
        if (finished != 0 && lane_in(finished)) p.counts[opix] = cnt <= p.total ? (int32_t)cnt : 0;
        // ---- deadline: lanes that ran mrd-1 steps without escaping -> 0 (only when the alarm is due)
        if ((int32_t)(n - bound) >= 0) {
            const bool is_live = lane_in(live);
            const uint32_t age = n - start;
            const bool expired = is_live && age >= p.total;
            if (expired) p.counts[opix] = 0;
            live &= ~__ballot(expired);
            uint32_t rem = (is_live && !expired) ? p.total - age : kFar;
            rem = wave_min_u32(rem < kFar ? rem : kFar);
            bound = n + uniform_u32(rem);
        }
        // ---- refill: free lanes take the next pixels of the current block (a second block if it runs out)
        const bool was_empty = (live == 0);
        while (more && ~live != 0) {
            if (blk_pos >= 64u) {
                if (blk_left > 0) {
                    ++blk;
                    --blk_left;
                } else {
                    uint32_t got;
                    blk = pop_blocks(wq, pp, p.nblocks, home, lane, p.batch, &got);
                    if (blk == kNoBlock) {
                        more = false;
                        break;
                    }
                    blk_left = got - 1u;
                }
                const uint32_t by = blk / p.bxn;
                blk_col = (blk - by * p.bxn) * 8u;
                blk_row = by * 8u;
                blk_pos = 0;
            }
            const unsigned long long free_lanes = ~live;
            const uint32_t navail = 64u - blk_pos;
            const uint32_t rank = rank_in(free_lanes);
            const bool take = lane_in(free_lanes) && rank < navail;
            if (take) {
                const uint32_t pidx = blk_pos + rank;
                const uint32_t lc = blk_col + (pidx & 7u), lr = blk_row + (pidx >> 3);
                cr = (double)(p.col0 + lc) * p.re_step + p.re_start;
                ci = (double)(p.row0 + lr) * p.im_step + p.im_start;
                zr = cr;
                zi = ci;
                a = zr * zr;
                b = zi * zi;
                start = n;
                cnt = 0;
                opix = lr * p.pitch + lc;
            }
            const unsigned long long taken = __ballot(take);
            live |= taken;
            blk_pos += (uint32_t)__popcll(taken);
        }
        if (live == 0) break;  // nothing in flight and (necessarily) nothing left to pop
        if (was_empty) bound = n + (p.total < kFar ? p.total : kFar);

        // ---- run the grouped hot loop until the next event
        live_in = live;
        {
            double t, pr, m, zr2, zi2, a2, b2, zrt, zit, at, bt;
            uint32_t k, k2, k3;
            unsigned long long save, tmp, tmp2, esc;
            uint32_t alarm = uniform_u32(bound);
            const uint32_t livemin = uniform_u32(more ? p.livemin : 0u);
            const uint32_t patience = uniform_u32(more ? p.patience : kFar);
            n = uniform_u32(n);
            live = uniform_u64(live);
            asm volatile(MBK_RFG_LOOP
                         : [zr] "+&v"(zr), [zi] "+&v"(zi), [a] "+&v"(a), [b] "+&v"(b), [cnt] "+&v"(cnt),
                           [zr2] "=&v"(zr2), [zi2] "=&v"(zi2), [a2] "=&v"(a2), [b2] "=&v"(b2),
                           [zrt] "=&v"(zrt), [zit] "=&v"(zit), [at] "=&v"(at), [bt] "=&v"(bt),
                           [t] "=&v"(t), [p] "=&v"(pr), [m] "=&v"(m), [n] "+&s"(n), [live] "+&s"(live),
                           [alarm] "+&s"(alarm), [k] "=&s"(k), [k2] "=&s"(k2), [k3] "=&s"(k3),
                           [save] "=&s"(save), [tmp] "=&s"(tmp), [tmp2] "=&s"(tmp2), [esc] "=&s"(esc)
                         : [cr] "v"(cr), [ci] "v"(ci), [start] "v"(start), [livemin] "s"(livemin),
                           [patience] "s"(patience)
                         : "vcc", "scc");
        }
    }
}

// counts -> quantised bytes for launches whose kernel wrote counts only
__global__ __launch_bounds__(256) void quantise_kernel(const int32_t *__restrict__ counts,
                                                       uint8_t *__restrict__ bytes, uint64_t n, int32_t mrd,
                                                       uint32_t wide, double rcp)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        bytes[i] = quantise(counts[i], mrd, wide, rcp);
}

}  // namespace mbk

// mbk_split.h -- the SPLIT of deep zooms (round 5): blocks in which every pixel stays inside run one wave per block on the
// units kernel (mbk_units.h: lane activity 1), the blocks with escaping pixels run on the lane-refill kernel (mbk_persist.h).
//
// Why (scripts/split_refill_model.py on the exact counts of BASELINE cfg3, 8192^2 at mrd 10 000; profiles/r05/
// split_refill_model.txt): 7 % of the blocks are all-alive and hold 60 % of the wave-steps at lane activity 1; the other 93 %
// hold 40 % of the wave-steps at lane activity 0.587 -- their pixels live 300 steps on average, a wave runs until its
// slowest lane (500).  Round 3's `refill` ran EVERYTHING through persistent waves: the all-alive blocks gained nothing and
// paid its 8-step groups, deadline checks and a drain of 10 000-step pixels in half-empty waves; the launch lost 8 %.  Split,
// the refill side holds short-lived pixels only (no long drain) and the long blocks keep the leaner loop: the model gives
// x1.09 .. x1.13 for the strict launch and x1.09 .. x1.14 with the cycle test (which stays on the units side; the few
// never-escaping pixels of the refill side run all their steps, each in a lane of its own).
//
// The classifier is the one the launch can afford: the centre pixel of every block for `probe` steps (512: 0.7 % of the
// launch's wave-steps, beside the previous launch on the auxiliary stream).  A block whose centre is still inside goes to the
// units list, everything else -- and only regular blocks: the refill kernel computes coordinates by the regular formula -- to
// the refill list.  A block filed on the units side although some of its pixels escape runs at its lock-step cost, as before;
// one filed on the refill side although it is all-alive costs its pixels a lane each.  Scheduling only: both kernels compute
// the reference's loop for every pixel they are given, and every block is in exactly one list.
#pragma once

#include "mbk_persist.h"
#include "mbk_units.h"

namespace mbk {

// Lists in the units kernel's buffer (mbk_units.h): units side = the H list, order[0 ..), count at order[n] (V and M counts at
// order[n+1], order[n+2] stay 0, the late-M / settled counts at order[2n+3], order[2n+4] too); refill side = order[n+3 ..), count
// at order[2n+5].
__global__ __launch_bounds__(1024) void classify_split_kernel(TileArgs p, uint32_t nregions, int32_t probe_steps, uint32_t *order)
{
    __shared__ uint32_t s_cnt[2][16], s_base[2];
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t cls = 2u;     // 0 units side, 1 refill side, 2 nothing
    uint32_t by = 0, bx = 0;
    if (r < nregions) {
        by = r / p.blocks_x;
        bx = r - by * p.blocks_x;
        uint32_t lc = bx * 8u + 4u, lr = by * 8u + 4u;
        lc = lc < p.ncols ? lc : p.ncols - 1u;
        lr = lr < p.nrows ? lr : p.nrows - 1u;
        const double cr = axis_value(p.re, p.col0 + lc), ci = axis_value(p.im, p.row0 + lr);
        const int32_t cap = p.mrd < probe_steps ? p.mrd : probe_steps;
        const int32_t cnt = cap > 1 ? escape_count<true>(cr, ci, cap) : 1;
        // whole 8x8 blocks away from the pinned end points only: the refill kernel has no form for anything else
        const bool regular = bx < p.fast_bx_end && by < p.fast_by_end && bx * 8u + 8u <= p.ncols && by * 8u + 8u <= p.nrows;
        cls = (cnt != 0 && regular) ? 1u : 0u;
    }
    unsigned long long m[2];
#pragma unroll
    for (uint32_t k = 0; k < 2u; ++k) {
        m[k] = __ballot(cls == k);
        if (lane == 0) s_cnt[k][wave] = (uint32_t)__popcll(m[k]);
    }
    __syncthreads();
    if (threadIdx.x < 2u) {
        const uint32_t k = threadIdx.x, nw = (blockDim.x + 63u) >> 6;
        uint32_t t = 0;
        for (uint32_t w = 0; w < nw; ++w) {
            const uint32_t c = s_cnt[k][w];
            s_cnt[k][w] = t;
            t += c;
        }
        s_base[k] = t ? atomicAdd(k == 0u ? &order[nregions] : &order[2u * nregions + 5u], t) : 0u;
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    if (cls == 0u) order[s_base[0] + s_cnt[0][wave] + (uint32_t)__popcll(m[0] & below)] = (by << 16) | bx;
    if (cls == 1u) order[nregions + 3u + s_base[1] + s_cnt[1][wave] + (uint32_t)__popcll(m[1] & below)] = (by << 16) | bx;
}

}  // namespace mbk

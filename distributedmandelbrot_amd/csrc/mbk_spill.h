// mbk_spill.h -- SPILL, second half (round 6; the first half and the why: block_pixel_spill in mbk_kernels.h).
//
// After the one-wave-per-block kernel: spill_cnt[b] = lanes block b handed over (0 for most blocks) | its checkpoint's number << 8,
// their states in the block's own slots b * T .. b * T + lanes - 1.  Three small kernels compact that into a list of slot numbers -- chunk
// sums (1024 blocks per workgroup), an exclusive scan of the chunk sums (rle_scan_kernel, one workgroup), the expansion -- and
// tile_spill_kernel runs the listed lanes 64 to a wave.  No device-scope atomics anywhere; the order of the list (checkpoint,
// then image order of the blocks) is a deterministic function of the window.
#pragma once

#include "mbk_kernels.h"

namespace mbk {

constexpr uint32_t kSpillChunk = 1024;   // blocks per workgroup of the two list kernels
constexpr uint32_t kSpillLevels = 12;    // checkpoints told apart in the list's order (a block's count word: lanes | level << 8)

// The list is ordered by checkpoint, LATEST FIRST, blocks in image order within one: a lane that has outlived more steps is
// likelier to outlive the rest (the pixels of the set end up in the late checkpoints), and the second pass lasts as long as
// its longest wave started late -- in image order alone it took 0.64 ms for 0.25 ms of work on cfg3 (profiles/r06/spill_ab.txt).
// chunk_sums[(kSpillLevels - 1 - level) * nchunks + chunk] = lanes of that level in that chunk; one scan over all of it.
__global__ __launch_bounds__(1024) void spill_count_kernel(const uint32_t *__restrict__ cnt, uint32_t nblocks, uint32_t nchunks, uint32_t *chunk_sums)
{
    __shared__ uint32_t s_lvl[kSpillLevels];
    if (threadIdx.x < kSpillLevels) s_lvl[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t b = blockIdx.x * kSpillChunk + threadIdx.x;
    const uint32_t w = b < nblocks ? cnt[b] : 0u;
    if (w & 0xffu) atomicAdd(&s_lvl[w >> 8], w & 0xffu);
    __syncthreads();
    if (threadIdx.x < kSpillLevels) chunk_sums[(kSpillLevels - 1u - threadIdx.x) * nchunks + blockIdx.x] = s_lvl[threadIdx.x];
}

// chunk_offsets: the exclusive scan of the chunk sums; src[j] = the slot of the j-th spilled lane of the launch
__global__ __launch_bounds__(1024) void spill_expand_kernel(const uint32_t *__restrict__ cnt, uint32_t nblocks, uint32_t nchunks,
                                                            const uint32_t *__restrict__ chunk_offsets, uint32_t lanes, uint32_t *src)
{
    __shared__ uint32_t s_off[16], s_present;
    const uint32_t b = blockIdx.x * kSpillChunk + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t w = b < nblocks ? cnt[b] : 0u, k = w & 0xffu, level = w >> 8;
    if (threadIdx.x == 0) s_present = 0u;
    __syncthreads();
    if (k) atomicOr(&s_present, 1u << level);
    __syncthreads();
    const uint32_t present = s_present;
    for (uint32_t l = 0; l < kSpillLevels; ++l) {
        if (!(present >> l & 1u)) continue;     // (uniform)
        const uint32_t v = (k && level == l) ? k : 0u;
        uint32_t incl = v;   // inclusive scan across the wave
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if (lane >= (uint32_t)off) incl += up;
        }
        __syncthreads();     // (s_off of the level before has been read)
        if (lane == 63u) s_off[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = chunk_offsets[(kSpillLevels - 1u - l) * nchunks + blockIdx.x];
            for (int q = 0; q < 16; ++q) {
                const uint32_t t = s_off[q];
                s_off[q] = run;
                run += t;
            }
        }
        __syncthreads();
        const uint32_t base = s_off[wave] + incl - v;
        for (uint32_t r = 0; r < v; ++r) src[base + r] = b * lanes + r;
    }
}

// The second pass: lane j of the launch continues the pixel behind src[j] from the state its block left.  The lanes of a wave
// come from different blocks and, with more than one checkpoint, have different numbers of steps behind them (n0): the loop
// counts steps RELATIVE to each lane's own start, runs until the lane with the most steps left is through (rel_total, uniform),
// and a lane's count is n0 + its relative escape step -- if that is a step the reference would still have run (<= mrd - 1),
// else 0: a lane that carries on beyond its own mrd - 1 only keeps the wave company.  8-step groups, no per-step prologue
// (these pixels outlived hundreds of steps).  Workgroup b takes the chunks b, b + nwg, ... of 64 lanes; the number of lanes is
// read from device memory (the scan's total), the grid is an upper bound.
template <typename T, bool kCycle>
__global__ __launch_bounds__(64) void tile_spill_kernel(TileArgs p, const uint32_t *__restrict__ src, const unsigned long long *total_ptr, uint32_t nwg)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t n_lanes = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)*total_ptr);
    const uint32_t total = p.mrd > 1 ? (uint32_t)p.mrd - 1u : 0u;
    for (uint32_t chunk = blockIdx.x; (unsigned long long)chunk * 64ull < n_lanes; chunk += nwg) {
        const uint32_t j = chunk * 64u + lane;
        const bool valid = j < n_lanes;
        const uint32_t slot = valid ? src[j] : 0u;
        const uint32_t meta = valid ? p.spill_meta[slot] : 0u;
        typename SpillPair<T>::type z;
        z.x = z.y = 0;
        if (valid) z = reinterpret_cast<const typename SpillPair<T>::type *>(p.spill_z)[slot];
        const uint32_t b = slot / p.spill_lanes, by = b / p.blocks_x, bx = b - by * p.blocks_x;
        const uint32_t lc = bx * 8u + (meta & 7u), lr = by * 8u + ((meta >> 3) & 7u), n0 = meta >> 6;
        // (only interior blocks spill: the regular coordinate formula, as in block_pixel_spill)
        const T cr = (T)((double)(p.col0 + lc) * p.re.step + p.re.start);
        const T ci = (T)((double)(p.row0 + lr) * p.im.step + p.im.start);
        T zr = z.x, zi = z.y, a = zr * zr, bb = zi * zi, m = 0;
        const uint32_t left = valid ? total - n0 : 0u;     // steps the reference still runs for this lane
        uint32_t rel_total = left;
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = __shfl_xor(rel_total, off, 64);
            rel_total = o > rel_total ? o : rel_total;
        }
        rel_total = (uint32_t)__builtin_amdgcn_readfirstlane((int)rel_total);
        int32_t cnt = 0;
        if (kCycle) {
            // the cycle test's first window: what the schedule of an unbroken run would stand at after the steps these lanes have
            // behind them (windows of ~ n / 4 steps = n / 32 checks), not 1 -- the orbits that got here have long periods or
            // settle slowly, short windows only make them wait (TileArgs::spill_win_shift; 31 = start at 1)
            uint32_t n0_min = valid ? n0 : 0xffffffffu;
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = __shfl_xor(n0_min, off, 64);
                n0_min = o < n0_min ? o : n0_min;
            }
            uint32_t win0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)n0_min) >> p.spill_win_shift;
            win0 = win0 ? win0 : 1u;
            if (valid) escape_steps_group<8, true, true, true>(cr, ci, zr, zi, a, bb, m, cnt, 0u, rel_total, p.cyc_window, win0);
        } else if (valid) {
            escape_steps_tail<8, false>(cr, ci, zr, zi, a, bb, m, cnt, 0u, rel_total);
        }
        if (valid) {
            const int32_t count = (cnt > 0 && (uint32_t)cnt <= left) ? (int32_t)(n0 + (uint32_t)cnt) : 0;
            const size_t o = (size_t)(lr + p.out_row0) * p.out_pitch + lc + p.out_col0;
            if (p.counts) p.counts[o] = count;
            if (p.bytes) p.bytes[o] = quantise(count, p);
        }
    }
}

}  // namespace mbk

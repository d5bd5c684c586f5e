// mbk_scan.h -- kernel "scan" (default since round 2): a persistent pass over all 8x8 blocks of a tile, then
// one workgroup per block that is still running.
//
// Why: with one single-wave workgroup per 8x8 block (kernels "asm"/"group") a 4096^2 tile is 262 144
// workgroups, and the workgroup dispatcher -- not the VALU, not HBM -- bounds every block whose pixels
// escape within a few steps (0.27 ns per workgroup: the all-exterior DataChunk (4,0,0) took 71 us against
// an HBM-write floor of ~15 us, and at level 16 of the reference's pyramid 3 tiles in 4 are of that kind).
// The light blocks need coarse, static scheduling (they all cost the same); the heavy ones -- whose cost
// varies 100x -- need exactly what the hardware dispatcher is good at.  So:
//
//   pass 1  tile_scan_kernel   a grid that fills the chip once; wave w takes blocks w, w+W, w+2W, ...
//           (W is a whole number of block rows whenever that fits: a wave stays in one block column for
//           `col_period` sweeps -- the real coordinate is computed once per column -- and then jumps to a
//           far column, so that no wave is stuck in the columns that cross the set).
//           For each block: the first `scan_steps` steps of the reference loop (WorkerCUDA.py:39-68) --
//           `exact_steps` with the per-step test, then whole grouped trips --, results written for
//           every pixel (0 for the ones still running: that IS their final value if they never
//           escape).  A block that still has running pixels is DEFERRED: its (zr, zi) state goes to
//           HBM (16 B per pixel) and its id + live-lane mask are appended to one of 64 lists (block id
//           mod 64; "dense" blocks -- every lane still running: interior of the set or of a slow
//           region -- from the front of the list's slab, the others from its back).
//   pass 2  tile_heavy_kernel  one single-wave workgroup per deferred block, dealt by the hardware
//           dispatcher: workgroup j takes element j div 64 of list j mod 64 (dense elements first, so
//           the long blocks start first), reloads the state and continues the grouped loop at step
//           `scan_steps`; lanes that escape overwrite their 0.  The host cannot know how many blocks
//           pass 1 deferred without a round trip, so the grid is 64 x (a hint): the longest list of the
//           PREVIOUS launch on the stream, which pass 2 itself writes to pinned host memory, plus
//           25 %; workgroups beyond a list's end leave after one load, and if the hint was too small a
//           workgroup simply continues with element j div 64 + hint, + 2 hint, ... of its list.
//           (Two persistent designs were measured first on cfg2, 510 us of work in this pass: popping
//           with one atomicAdd per block took 770 us -- the waves of a SIMD finish equal blocks in
//           lock-step, 128 of them hit each cursor at once, 29 us per pop; a static deal took 563 us
//           but lost 35 % on the deep zoom cfg3, whose blocks all look alike after 24 steps and then
//           run for 25 ... 9999.)
//
// Bit-exactness: every pixel runs exactly the arithmetic of the "group" kernel (same loops from
// mbk_loops.inc, same state, only split at a step boundary and moved through HBM as raw bits); waves
// touching the |c| = 2 ring finish in pass 1 with the per-step loop.  Nothing in the results depends on
// which wave processes which block, nor on the hint.
//
// Scratch (per stream, mbk_api.hip): entries 16 B per block, state 2*sizeof(T) B per pixel (worst case:
// every block deferred), two cursor sets used alternately -- pass 1 of launch L clears the set of launch
// L+1, so no memset sits between launches (launches on one stream are ordered) -- and the 4-byte hint.
#pragma once

#include "mbk_refill.h"  // lane_in, uniform_u32/u64 (includes mbk_kernels.h)

namespace mbk {

constexpr uint32_t kScanQueues = 64;

struct ScanCursors {  // one 64-byte line per list: lengths, appended to by pass 1 (atomicAdd), read-only in pass 2
    struct Q {
        unsigned int tail_dense, tail_sparse, pad0[14];
    } q[kScanQueues];
};

struct ScanEntry {
    uint32_t block;
    uint32_t pad;
    unsigned long long live;  // lanes still running after pass 1
};

struct ScanArgs {
    ScanCursors *cur;       // this launch's cursors (all zero on entry)
    ScanCursors *cur_next;  // cleared by pass 1 for the next launch on this stream
    ScanEntry *entries;     // kScanQueues * qcap
    void *state;            // kScanQueues * qcap * 64 * {T zr, T zi}
    uint32_t *hint_out;     // pinned host memory, 2 words written by pass 2: the longest list's length (next
                            // launch's grid hint) and the deferred share of all blocks x 65536 (kernel choice)
    uint32_t qcap;          // entries per list = ceil(nblocks / 64)
    uint32_t nblocks;
    uint32_t scan_steps;    // pass 1 depth: exact_steps + a multiple of 16
    // Grid sizes as explicit arguments: gridDim.x lives in the dispatch packet, and the compiler re-read
    // it on every trip of the block loop -- a host-memory access per block, ~3 us each: pass 1 of an
    // all-exterior tile took 110 us instead of 37.
    uint32_t stride;        // pass 1: wave w takes blocks w, w + stride, ...
    uint32_t stride_bx;     // stride mod blocks_x (0: whole block rows per sweep)
    uint32_t stride_by;     // stride div blocks_x
    uint32_t col_period;    // pass 1, stride_bx == 0: sweeps a wave stays in one block column (0 = for ever)
    uint32_t col_jump;      // ... then it moves this many block columns to the right (mod blocks_x)
    uint32_t xcd_map;       // pass 1: 1 = XCD-aware block-column permutation (needs blocks_x % 32 == 0, stride_bx == 0)
    uint32_t ranks2;        // pass 2: elements per list covered by the grid (grid = 64 * ranks2)
    uint32_t long_groups;   // pass 2: 1 = dense blocks use 16-step groups (option group_steps == 16)
};

template <typename T>
struct ScanState {
    T zr, zi;
};

// np.linspace sample k for the views this kernel accepts (step != 0; the host sends the others to the
// "group" kernel): fl(fl(k*step) + start), end point pinned.  Same arithmetic as axis_value.
__device__ __forceinline__ double scan_axis_value(const Axis &a, uint32_t k)
{
    const double v = (double)k * a.step + a.start;
    return (k + 1u == a.n) ? a.last : v;
}

template <typename T>
__device__ __forceinline__ void store_results(const TileArgs &p, size_t o, int32_t count, T m)
{
    if (p.counts) p.counts[o] = count;
    if (p.bytes) p.bytes[o] = quantise(count, p);
    if (p.smooth) p.smooth[o] = smooth_value(count, (double)m);
}

template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(94))) void tile_scan_kernel(TileArgs p, ScanArgs s)
{
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x == 0) {  // clear the next launch's cursors
        unsigned int *w = reinterpret_cast<unsigned int *>(s.cur_next);
        for (uint32_t k = lane; k < (uint32_t)(sizeof(ScanCursors) / 4u); k += 64u) w[k] = 0u;
    }
    const uint32_t total = p.mrd > 1 ? (uint32_t)p.mrd - 1u : 0u;
    const uint32_t first = total < p.exact_steps ? total : p.exact_steps;
    const uint32_t depth = total < s.scan_steps ? total : s.scan_steps;
    const T margin = sizeof(T) == 8 ? (T)1e-9 : (T)1e-3;
    ScanState<T> *state = static_cast<ScanState<T> *>(s.state);
    const bool use_prologue = p.smooth == nullptr && depth >= 4u;   // (depth >= 4: pass 1 runs at least 4 steps)

    // Block coordinates are kept incrementally: one sweep is (bx, by) -> (bx + stride_bx, by + stride_by).
    // XCD-aware start (s.xcd_map): the hardware deals consecutive workgroup ids to the 8 XCDs in turn, each
    // with its own L2.  With the identity mapping the four 8-pixel-wide blocks that share a 128-byte line of
    // the output are written by four different XCDs and reach HBM as four partial lines; the bit
    // permutation below gives each XCD runs of four adjacent block columns, so a line is completed in ONE L2.
    // (col_jump is a multiple of 32 columns, which keeps that property.)
    uint32_t by = blockIdx.x / p.blocks_x, bx = blockIdx.x - by * p.blocks_x;
    if (s.xcd_map) {
        const uint32_t a = bx >> 3, c = bx & 7u;         // bx = 8a + c, c = the XCD this wave runs on
        bx = ((a >> 2) << 5) | (c << 2) | (a & 3u);
    }
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    uint32_t lc = bx * 8u + lx;
    T cr = (T)scan_axis_value(p.re, p.col0 + lc);
    uint32_t sweeps_here = 0;
    const uint32_t nby = (p.nrows + 7u) / 8u;
    for (; by < nby; by += s.stride_by) {
        // (stride_bx != 0 -- a grid that is not a whole number of block rows -- only happens for windows
        // narrower than the chip or wider than 65536 pixels; then by may step past the last row by one)
        const uint32_t b = by * p.blocks_x + bx;
        if (b >= s.nblocks) break;
        const uint32_t lr = by * 8u + ly;
        const bool valid = lc < p.ncols && lr < p.nrows;
        bool running = false;
        T zr = 0, zi = 0;
        if (valid) {
            const T ci = (T)scan_axis_value(p.im, p.row0 + lr);
            T a, bq, m = 0;
            int32_t cnt = 0;
            zr = cr;
            zi = ci;
            a = zr * zr;
            bq = zi * zi;
            // waves touching the |c| = 2 ring: per-step loop to the end (see tile_asm_kernel)
            bool risky = false;
            if (p.ring_possible) {
                const T c2 = a + bq;
                risky = __any(c2 > (T)4 - margin && c2 < (T)4 + margin) != 0;
            }
            if (risky) {
                escape_steps_asm<true>(cr, ci, zr, zi, a, bq, m, cnt, 0u, total);
            } else {
                uint32_t done = 0u;   // steps already taken by the lanes still running
                bool inside = true;
                if (use_prologue) {   // wave-uniform: 4 branch-free steps, see escape_steps_prologue4
                    escape_steps_prologue4(cr, ci, zr, zi, a, bq, cnt);
                    inside = cnt == 5;
                    cnt = inside ? 0 : cnt;
                    done = 4u;
                }
                if (inside) {
                    if (first > done) escape_steps_asm<true>(cr, ci, zr, zi, a, bq, m, cnt, done, first);
                    const uint32_t from = first > done ? first : done;
                    if (cnt == 0 && depth > from) escape_steps_group<8>(cr, ci, zr, zi, a, bq, m, cnt, from, depth);
                    running = cnt == 0 && total > depth;
                }
            }
            store_results<T>(p, (size_t)(lr + p.out_row0) * p.out_pitch + lc + p.out_col0, cnt, m);
        }
        const unsigned long long live = __ballot(running);
        if (live != 0ull) {  // wave-uniform: defer the block
            const uint32_t q = b & (kScanQueues - 1u);
            const bool dense = live == ~0ull;
            uint32_t idx = 0;
            if (lane == 0) idx = atomicAdd(dense ? &s.cur->q[q].tail_dense : &s.cur->q[q].tail_sparse, 1u);
            idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
            const uint32_t e = q * s.qcap + (dense ? idx : s.qcap - 1u - idx);
            if (lane == 0) {
                s.entries[e].block = b;
                s.entries[e].live = live;
            }
            if (running) {
                ScanState<T> st;
                st.zr = zr;
                st.zi = zi;
                state[(size_t)e * 64u + lane] = st;
            }
        }
        // next block of this wave: same column, stride_by rows down -- except every col_period sweeps
        // (or on every sweep when the grid is not a whole number of rows)
        bool moved = false;
        if (s.stride_bx != 0u) {
            bx += s.stride_bx;
            if (bx >= p.blocks_x) {
                bx -= p.blocks_x;
                by += 1u;
            }
            moved = true;
        } else if (s.col_period != 0u && ++sweeps_here == s.col_period) {
            sweeps_here = 0u;
            bx += s.col_jump;
            if (bx >= p.blocks_x) bx -= p.blocks_x;
            moved = true;
        }
        if (moved) {  // wave-uniform
            lc = bx * 8u + lx;
            cr = (T)scan_axis_value(p.re, p.col0 + lc);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(64) void tile_heavy_kernel(TileArgs p, ScanArgs s)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t q = blockIdx.x & (kScanQueues - 1u);
    uint32_t i = blockIdx.x >> 6;
    const uint32_t len_d = s.cur->q[q].tail_dense, len = len_d + s.cur->q[q].tail_sparse;
    if (blockIdx.x == 0) {  // next launch's hints (posted writes to pinned host memory)
        uint32_t n = s.cur->q[lane].tail_dense + s.cur->q[lane].tail_sparse, sum = n;
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)n, off, 64);
            n = o > n ? o : n;
            sum += (uint32_t)__shfl_xor((int)sum, off, 64);
        }
        if (lane == 0) {
            s.hint_out[0] = n;
            s.hint_out[1] = (uint32_t)(((uint64_t)sum << 16) / s.nblocks);
        }
    }
    const uint32_t total = p.mrd > 1 ? (uint32_t)p.mrd - 1u : 0u;
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const ScanState<T> *state = static_cast<const ScanState<T> *>(s.state);
    for (; i < len; i += s.ranks2) {  // one trip when the hint was large enough
        const uint32_t e = q * s.qcap + (i < len_d ? i : s.qcap - 1u - (i - len_d));
        const uint32_t blk = uniform_u32(s.entries[e].block);
        const unsigned long long live = uniform_u64(s.entries[e].live);
        const uint32_t by = blk / p.blocks_x, bx = blk - by * p.blocks_x;
        const uint32_t lc = bx * 8u + lx, lr = by * 8u + ly;
        if (lane_in(live)) {
            const ScanState<T> st = state[(size_t)e * 64u + lane];
            const T cr = (T)scan_axis_value(p.re, p.col0 + lc);
            const T ci = (T)scan_axis_value(p.im, p.row0 + lr);
            T zr = st.zr, zi = st.zi, a = zr * zr, bq = zi * zi, m = 0;
            int32_t cnt = 0;
            // dense blocks (interior): 16-step groups, 6.125 slots per step; the others: 8 (cheaper replays)
            if (s.long_groups && i < len_d) escape_steps_tail<16>(cr, ci, zr, zi, a, bq, m, cnt, s.scan_steps, total);
            else escape_steps_group<8>(cr, ci, zr, zi, a, bq, m, cnt, s.scan_steps, total);
            if (cnt > 0) store_results<T>(p, (size_t)(lr + p.out_row0) * p.out_pitch + lc + p.out_col0, cnt, m);
        }
    }
}

}  // namespace mbk

// mbk_scan.h -- kernel "scan" (default since round 2): two persistent passes over the 8x8 blocks of a tile.
//
// Why: with one single-wave workgroup per 8x8 block (kernels "asm"/"group") a 4096^2 tile is 262 144
// workgroups, and the workgroup dispatcher -- not the VALU, not HBM -- bounds every block whose pixels
// escape within a few steps (0.27 ns per workgroup: the all-exterior DataChunk (4,0,0) took 71 us against
// an HBM-write floor of ~15 us, and at level 16 of the reference's pyramid 3 tiles in 4 are of that kind).
// The light blocks need coarse, static scheduling (they all cost the same); the heavy ones need fine,
// dynamic scheduling (their cost varies 100x).  So:
//
//   pass 1  tile_scan_kernel   a grid that fills the chip once; wave w takes blocks w, w+W, w+2W, ...
//           For each block: coordinates, the first `scan_steps` steps of the reference loop
//           (WorkerCUDA.py:39-68) -- `exact_steps` with the per-step test, then whole grouped trips --,
//           results written for every pixel (0 for the ones still running: that IS their final value
//           if they never escape).  A block that still has running pixels is DEFERRED: its (zr, zi)
//           state goes to HBM (16 B per pixel) and its id + live-lane mask are appended to one of 64
//           queues (block id mod 64; "dense" blocks -- every lane still running, i.e. interior of the
//           set or of a slow region -- from the front, the others from the back).
//   pass 2  tile_heavy_kernel  persistent waves pop deferred blocks (dense ones first: longest jobs
//           first), reload the state and continue the grouped loop at step `scan_steps`; lanes that
//           escape overwrite their 0.
//
// Bit-exactness: every pixel runs exactly the arithmetic of the "group" kernel (same loops from
// mbk_loops.inc, same state, only split at a step boundary and moved through HBM as raw bits); waves
// touching the |c| = 2 ring finish in pass 1 with the per-step loop.  Nothing in the results depends on
// the queueing order.
//
// Scratch (per stream, mbk_api.hip): entries 16 B per block, state 2*sizeof(T) B per pixel (worst case:
// every block deferred), two cursor sets used alternately -- pass 1 of launch L clears the set of launch
// L+1, so no memset sits between launches (launches on one stream are ordered).
#pragma once

#include "mbk_refill.h"  // lane_in, uniform_u32/u64 (includes mbk_kernels.h)

namespace mbk {

constexpr uint32_t kScanQueues = 64;

struct ScanCursors {  // one 64-byte line per counter pair
    struct Q {
        unsigned int tail_dense, tail_sparse, pad0[14];  // appended by pass 1
        unsigned int head_dense, head_sparse, pad1[14];  // popped by pass 2
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
    uint32_t qcap;          // entries per queue = ceil(nblocks / 64)
    uint32_t nblocks;
    uint32_t scan_steps;    // pass 1 depth: exact_steps + a multiple of 16
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

// Lane coordinates of block b (8x8 pixels, one per lane).
struct BlockPos {
    uint32_t lc, lr;  // column / row inside the window
    bool valid;
};
__device__ __forceinline__ BlockPos block_pos(const TileArgs &p, uint32_t b, uint32_t lane)
{
    const uint32_t by = b / p.blocks_x, bx = b - by * p.blocks_x;
    BlockPos r;
    r.lc = bx * 8u + (lane & 7u);
    r.lr = by * 8u + (lane >> 3);
    r.valid = r.lc < p.ncols && r.lr < p.nrows;
    return r;
}

template <typename T>
__device__ __forceinline__ void store_results(const TileArgs &p, const BlockPos &pos, int32_t count, T m)
{
    const size_t o = (size_t)(pos.lr + p.out_row0) * p.out_pitch + pos.lc + p.out_col0;
    if (p.counts) p.counts[o] = count;
    if (p.bytes) p.bytes[o] = quantise(count, p);
    if (p.smooth) p.smooth[o] = smooth_value(count, (double)m);
}

template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(94))) void tile_scan_kernel(TileArgs p, ScanArgs s)
{
    const uint32_t lane = threadIdx.x;
    if (blockIdx.x == 0) {  // clear the next launch's cursors (128 B per queue = 32 words)
        unsigned int *w = reinterpret_cast<unsigned int *>(s.cur_next);
        for (uint32_t k = lane; k < (uint32_t)(sizeof(ScanCursors) / 4u); k += 64u) w[k] = 0u;
    }
    const uint32_t total = p.mrd > 1 ? (uint32_t)p.mrd - 1u : 0u;
    const uint32_t first = total < p.exact_steps ? total : p.exact_steps;
    const uint32_t depth = total < s.scan_steps ? total : s.scan_steps;
    const T margin = sizeof(T) == 8 ? (T)1e-9 : (T)1e-3;
    ScanState<T> *state = static_cast<ScanState<T> *>(s.state);

    for (uint32_t b = blockIdx.x; b < s.nblocks; b += gridDim.x) {
        const BlockPos pos = block_pos(p, b, lane);
        bool running = false;
        T zr = 0, zi = 0;
        if (pos.valid) {
            const T cr = (T)scan_axis_value(p.re, p.col0 + pos.lc);
            const T ci = (T)scan_axis_value(p.im, p.row0 + pos.lr);
            // waves touching the |c| = 2 ring: per-step loop to the end (see tile_asm_kernel)
            T a, bq, m = 0;
            int32_t cnt = 0;
            zr = cr;
            zi = ci;
            a = zr * zr;
            bq = zi * zi;
            bool risky = false;
            if (p.ring_possible) {
                const T c2 = a + bq;
                risky = __any(c2 > (T)4 - margin && c2 < (T)4 + margin) != 0;
            }
            if (risky) {
                escape_steps_asm<true>(cr, ci, zr, zi, a, bq, m, cnt, 0u, total);
            } else {
                escape_steps_asm<true>(cr, ci, zr, zi, a, bq, m, cnt, 0u, first);
                if (cnt == 0 && depth > first) escape_steps_group<8>(cr, ci, zr, zi, a, bq, m, cnt, first, depth);
                running = cnt == 0 && total > depth;
            }
            store_results<T>(p, pos, cnt, m);
        }
        const unsigned long long live = __ballot(running);
        if (live != 0) {  // wave-uniform: defer the block
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
    }
}

// Pop one deferred block: dense class first, then sparse.  Returns the entry index or 0xffffffff.
// One atomicAdd per pop on the wave's current queue; when it is dry every lane looks at one queue and
// a ballot gives the non-empty set (same scheme as mbk_refill.h: no hot global word).
struct ScanPopper {
    uint32_t cq;       // current queue
    uint32_t cq_tail;  // its tail in the current class
    uint32_t sparse;   // 0: dense class, 1: sparse class
};

__device__ __forceinline__ uint32_t scan_pop(const ScanArgs &s, ScanPopper &pp, uint32_t home, uint32_t lane)
{
    for (;;) {
        uint32_t idx = 0;
        if (lane == 0)
            idx = atomicAdd(pp.sparse ? &s.cur->q[pp.cq].head_sparse : &s.cur->q[pp.cq].head_dense, 1u);
        idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)idx);
        if (idx < pp.cq_tail) return pp.cq * s.qcap + (pp.sparse ? s.qcap - 1u - idx : idx);
        // dry: every lane inspects one queue of the current class
        const ScanCursors::Q &mine = s.cur->q[lane];
        const uint32_t h = __hip_atomic_load(pp.sparse ? &mine.head_sparse : &mine.head_dense, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t t = pp.sparse ? mine.tail_sparse : mine.tail_dense;
        const unsigned long long m = __ballot(h < t);
        if (m == 0) {
            if (pp.sparse) return 0xffffffffu;
            pp.sparse = 1u;
            pp.cq = home;
            pp.cq_tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.cur->q[home].tail_sparse);
            continue;
        }
        const unsigned long long rot = (m >> home) | (m << ((64u - home) & 63u));
        pp.cq = (home + (uint32_t)__ffsll((long long)rot) - 1u) & 63u;
        const ScanCursors::Q &cqr = s.cur->q[pp.cq];
        pp.cq_tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)(pp.sparse ? cqr.tail_sparse : cqr.tail_dense));
    }
}

template <typename T>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(94))) void tile_heavy_kernel(TileArgs p, ScanArgs s)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t home = blockIdx.x & (kScanQueues - 1u);
    const uint32_t total = p.mrd > 1 ? (uint32_t)p.mrd - 1u : 0u;
    const ScanState<T> *state = static_cast<const ScanState<T> *>(s.state);
    ScanPopper pp;
    pp.cq = home;
    pp.sparse = 0u;
    pp.cq_tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.cur->q[home].tail_dense);
    for (;;) {
        const uint32_t e = scan_pop(s, pp, home, lane);
        if (e == 0xffffffffu) return;
        const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.entries[e].block);
        const unsigned long long live = uniform_u64(s.entries[e].live);
        const BlockPos pos = block_pos(p, b, lane);
        if (lane_in(live)) {
            const T cr = (T)scan_axis_value(p.re, p.col0 + pos.lc);
            const T ci = (T)scan_axis_value(p.im, p.row0 + pos.lr);
            const ScanState<T> st = state[(size_t)e * 64u + lane];
            T zr = st.zr, zi = st.zi, a = zr * zr, bq = zi * zi, m = 0;
            int32_t cnt = 0;
            escape_steps_group<8>(cr, ci, zr, zi, a, bq, m, cnt, s.scan_steps, total);
            if (cnt > 0) store_results<T>(p, pos, cnt, m);
        }
    }
}

}  // namespace mbk

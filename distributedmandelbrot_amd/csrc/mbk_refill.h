// mbk_refill.h -- building blocks of the persistent lane-refill kernel (mbk_persist.h):
//   * WorkQueues / pop_blocks: 64 block cursors in HBM (one atomicAdd per pop on the wave's current
//     queue -- 64 addresses, ~128 customers each; when that queue is dry every lane loads one cursor
//     and the ballot of "cursor < end" is a fresh non-empty mask, so the remaining work is found in
//     O(1) without a hot global word).  A single "non-empty mask" word read on every pop was measured
//     to serialise the whole chip (2.6 ms per cfg2 tile): 8192 waves hitting one address.
//   * MBK_RFG_LOOP: the grouped hot loop with a wave-uniform clock and relative counts.
//   * small wave-level helpers.
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

// Element i of a list that an EARLIER kernel wrote and this launch only reads (dispatch lists, XCD shares), with a scalar load
// wherever the code stands: the compiler turns an ordinary uniform load that follows the wave's own stores into a vector
// load plus readfirstlane (several hundred ns more per unit), and is free to keep "uniform" values in vector registers.
__device__ __forceinline__ uint32_t scalar_load_u32(const uint32_t *list, uint32_t i)
{
    uint32_t v;
    const uint32_t *a = list + i;   // (uniform by construction; a readfirstlane here makes the compiler park it in a VGPR)
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(a) : "memory");
    return v;
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

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
#define MBK_RFG_REPLAY_STEP(STEP, J, ID)                   \
    STEP                                                   \
    "v_add_f64 %[m], %[at], %[bt]\n"                       \
    "v_cmp_le_f64 vcc, 4.0, %[m]\n"                        \
    "s_add_u32 %[k], %[n], " J "\n"                        \
    "s_or_b64 %[esc], %[esc], vcc\n"                       \
    "s_and_saveexec_b64 %[tmp2], vcc\n"                    \
    "v_sub_u32 %[cnt], %[k], %[start]\n"                   \
    "s_andn2_b64 exec, %[tmp2], vcc\n"                     \
    "s_cbranch_scc0 .Lqrepend" ID "_%=\n"   /* every tripped lane found: stop replaying */
#define MBK_RFG_REPLAY8(FIRST, ID, J1, J2, J3, J4, J5, J6, J7, J8, NADV, COPYBACK) \
    ".Lqrep" ID "_%=:\n"                                   \
    "s_and_saveexec_b64 %[tmp], vcc\n"                     \
    "s_mov_b64 %[esc], 0\n"                                \
    MBK_RFG_REPLAY_STEP(FIRST, J1, ID) MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J2, ID) \
    MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J3, ID) MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J4, ID) \
    MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J5, ID) MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J6, ID) \
    MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J7, ID) MBK_RFG_REPLAY_STEP(MBK_RFG_T2T, J8, ID) \
    ".Lqrepend" ID "_%=:\n"                                \
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

}  // namespace mbk

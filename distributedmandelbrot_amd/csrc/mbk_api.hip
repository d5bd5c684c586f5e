// mbk_api.hip -- host side of libmbk_hip.so: the C ABI declared in include/mbk.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see build.py).
// -ffp-contract=off covers the HOST arithmetic in this file too (axis preparation, tile geometry):
// every fp64 operation below must round individually, as CPython / numpy do for the reference
// (DistributedMandelbrotWorkerCUDA.py:24-32,75-78).
#include "../../include/mbk.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mbk_kernels.h"
#include "mbk_refill.h"
#include "mbk_persist.h"
#include "mbk_scan.h"
#include "mbk_units.h"
#include "mbk_spill.h"
#include "mbk_feeder.h"

using mbk::Axis;
using mbk::ReduceOut;
using mbk::ReduceSlot;
using mbk::TileArgs;
using mbk::WorkQueues;

// Scratch that helper kernels of a launch use (dispatch-order list, work-queue cursors, the scan
// kernel's deferred-block lists).  It is keyed by the HIP stream the launch goes to: launches on one
// stream are ordered, so one set per stream can never be rewritten while a kernel still reads it,
// however many streams the caller uses (round 1 shared a ring of 8 across all streams).
// Dispatch lists per stream, used in turn.  Two were enough to let the pre-pass of launch L + 1 run beside the tile kernel of
// launch L -- but a second queue gets no wave slot on a chip full of single-wave workgroups until the tile kernel's grid is
// dealt (measured, profiles/r05/gaps_*.txt: the pre-pass starts 200 us into a 275 us launch and ends after it), which did not
// matter while a launch ended in a 20 us drain and sits on the critical path now that it does not.  With three the pre-pass
// of launch L + 1 may start one launch earlier and is done long before its list is needed.
static const int kOrderRing = 3;

struct StreamScratch {
    hipStream_t stream = nullptr;
    // heavy-first dispatch order of kernels "asm"/"group": kOrderRing lists (list | 3 counters | middle-class list) used
    // in turn, so that the pre-pass of a later launch (memset + classify, on the aux stream) can run while the tile
    // kernels of the launches before it still read the others
    uint32_t *d_order[kOrderRing] = {};
    uint32_t *d_ctl = nullptr;    // 8 words per list: the units pre-pass' counters (H, V, M | late M, settled H, ticket), zero between launches
    bool ctl_dirty = true;        // the counters must be cleared (on the pre-pass' own stream) before the next pre-pass: never used
                                  // yet, or a launch failed after its pre-pass was enqueued and may have left a ticket behind
    size_t order_cap = 0;         // regions
    unsigned order_turn = 0;
    hipStream_t aux = nullptr;    // the pre-pass stream
    uint32_t aux_prio = 0;        // 1 default priority, 2 highest (MBK_OPT_PREPASS_OVERLAP = 2)
    int pp_last = -1;             // MBK_OPT_PREPASS_OVERLAP of the last ordered launch on this stream (a change drains both streams)
    hipEvent_t ev_cls[kOrderRing] = {};   // pre-pass into list k finished
    hipEvent_t ev_done[kOrderRing] = {};  // the tile kernel that read list k finished
    bool done_valid[kOrderRing] = {};
    WorkQueues *d_queues = nullptr;  // kernel "refill"
    mbk::ScanCursors *d_cursors = nullptr;  // kernel "scan": two sets, used alternately
    uint32_t *d_entries = nullptr;   // kernel "scan": the 64 todo lists (block ids)
    size_t scan_cap_blocks = 0;      // their capacity, in ids
    unsigned scan_turn = 0;
    bool cursors_dirty = false;      // a scan launch failed after its cursor sets were assigned: clear both next time
    uint32_t *h_hint = nullptr;   // pinned, written by the kernels of the last launch on this stream:
                                  // [0] longest deferred list (scan pass 2), [1] share of heavy blocks x 65536
    // SPILL (mbk_spill.h): the slots of the spilled lanes (state, meta), the per-block counts, the chunk sums / offsets and the
    // total of the prefix sum, the compacted list of slots
    void *d_spill_z = nullptr;
    uint32_t *d_spill_meta = nullptr, *d_spill_cnt = nullptr, *d_spill_chunks = nullptr, *d_spill_src = nullptr;
    unsigned long long *d_spill_total = nullptr;
    size_t spill_cap_slots = 0, spill_cap_blocks = 0;   // capacities (slots of 16 bytes of state; blocks)
    ReduceSlot *d_red = nullptr;  // mbk_reduce_counts on this (caller) stream: its own partial results, so that a
    ReduceSlot *h_red = nullptr;  // reduction on a caller stream never shares a buffer with a tile in flight on a slot
    // kernel "units": the shares of the eight XCDs (mbk_units.h).  Launch number c (1-based) writes its time stamps into
    // slot c % kStampSlots of h_stamps (pinned) and used the fractions in xcd_ring[c % kShareRing].
    unsigned long long *h_stamps = nullptr;
    double xcd_f[8] = {0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125};
    uint32_t xcd_issued = 0, xcd_consumed = 0;
    struct SharesUsed { uint32_t seq, switches; float f[8]; } xcd_ring[64] = {};   // switches: the ctx' stream_switches at the launch
};
static const size_t kMaxStreamScratch = 64;
static const uint32_t kStampSlots = 16, kShareRing = 64;

// One in-flight tile of the host-buffer API: its own stream (so that the D2H of one slot overlaps the
// kernel of the other), events, device result buffers and reduction scratch.
struct Slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev_k0 = nullptr, ev_k1 = nullptr, ev_c0 = nullptr, ev_c1 = nullptr;
    int32_t *d_counts = nullptr;
    uint8_t *d_bytes = nullptr;
    size_t cap_px = 0;
    ReduceSlot *d_red = nullptr;   // mbk::kReduceSlots partial results, folded on the host (reduce_total)
    ReduceSlot *h_red = nullptr;  // pinned copy of d_red
    bool busy = false;           // submitted, not yet waited for
    bool with_bytes = false;
    uint8_t *lazy_h_bytes = nullptr;  // MBK_LAZY_UNIFORM without a copy enqueued at submit: the byte copy is decided in mbk_wait
    size_t lazy_px = 0;
    bool immediate = false;      // MBK_LAZY_UNIFORM, window wholly outside |c| = 2: nothing was enqueued, `imm` is the result
    mbk_stats imm = {};
};

struct mbk_ctx {
    int device = -1;
    Slot s[MBK_SLOTS];
    std::vector<StreamScratch> scratch;  // one per stream seen by this ctx
    size_t last_px = 0;              // pixels of the last tile computed with bytes (for mbk_serialize_last)
    double *d_smooth = nullptr;      // smooth-colouring output of the synchronous API
    size_t smooth_cap_px = 0;
    uint8_t *d_rle = nullptr;        // RLE scratch: block counts | run starts | run values | output stream
    size_t rle_cap_px = 0;
    uint32_t opt[MBK_OPT_COUNT_];    // tuning options (mbk_set_option); every value is bit-exact
    hipStream_t last_tile_stream = nullptr;   // launch_tile: the stream of the last tile launch, and how often it changed
    uint32_t stream_switches = 0;             // (xcd_shares_update: which launches had the chip to themselves)
    int scan_occ[2][2] = {{0, 0}, {0, 0}};  // resident single-wave workgroups per CU: [f64|f32][scan|heavy]
    int scan_occ_inline[2] = {0, 0};        // the same for pass 1 in its finish-in-place form (MBK_OPT_SCAN_INLINE)
    uint32_t wave_limit_lds[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // MBK_OPT_WAVE_LIMIT: dynamic LDS bytes per single-wave workgroup
                                            // that leave room for 4 x k workgroups per CU (k = index; 0 = no padding)
    // the host probe of the last window (window_heavy_share): one submit asks for it up to three times -- the copy decision of
    // MBK_LAZY_UNIFORM, the kernel choice, the units rule -- and all three must rest on the same evaluation
    uint32_t spill_launches = 0;              // launches that ran with the SPILL second pass (MBK_INFO_SPILL)
    hipStream_t last_spill_stream = nullptr;  // ... and the stream of the last one
    struct ProbeKey { Axis re, im; uint32_t col0, row0, ncols, nrows; } probe_key = {};
    double probe_share = 0.0;
    bool probe_valid = false;
    hipDeviceProp_t prop;
    std::string err;
};

static thread_local std::string g_err;

static int fail(mbk_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg;
    g_err = msg;
    return code;
}

#define MBK_HIP(ctx, call)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            return fail((ctx), MBK_ERR_HIP,                                                 \
                        std::string(#call) + ": " + hipGetErrorString(e_));                 \
        }                                                                                   \
    } while (0)

// ---- host arithmetic (individually rounded; `volatile` keeps the compiler honest at any -O) ----

static Axis make_axis(double start, double range, uint32_t n)
{
    Axis a;
    std::memset(&a, 0, sizeof(a));
    a.start = start;
    a.n = n;
    volatile double stop = start + range;
    if (n <= 1) {
        a.last = start;
        a.div = 1.0;
        return a;
    }
    volatile double delta = stop - start;
    volatile double div = (double)(n - 1);
    volatile double step = delta / div;
    a.last = stop;
    a.delta = delta;
    a.div = div;
    a.step = step;
    a.step_is_zero = (step == 0.0) ? 1u : 0u;
    return a;
}

static double axis_value_host(const Axis &a, uint32_t k)
{
    if (k + 1u == a.n) return a.last;
    volatile double y;
    if (a.step_is_zero) {
        volatile double q = (double)k / a.div;
        y = q * a.delta;
    } else {
        y = (double)k * a.step;
    }
    volatile double v = y + a.start;
    return v;
}

// Is the pinned last sample of the axis what the regular formula gives anyway, fl(fl((n-1)*step)+start) == stop?
// (np.linspace overwrites y[n-1] with `stop`; for every DataChunk level probed -- and for cfg2's axes -- the
// formula lands on it exactly.)  Then the blocks holding that sample need no special case in the fast paths.
static bool axis_end_is_regular(const Axis &a)
{
    if (a.n <= 1 || a.step_is_zero) return false;
    volatile double y = (double)(a.n - 1u) * a.step;
    volatile double v = y + a.start;
    return v == a.last;
}

// Coordinates must stay far from overflow so that no inf-inf = NaN can appear before the bailout
// test fires (the hand-scheduled kernels compare the high word of |z|^2 as an integer).
static const double kMaxCoord = 0x1p500;
// fma(2, zr*zi, ci) == fl(fl((2*zr)*zi) + ci) unless zr*zi is subnormal AND ci is so small that
// the last bit of a subnormal survives the addition.  |ci| >= 2^-900 (or ci == 0, where zi stays
// exactly 0) rules that out; otherwise use the literal (2*zr)*zi instantiation.
static const double kSafeImagMin = 0x1p-900;

// A multiplier coprime to n near n * (golden ratio - 1): id -> (id * m) mod n is a bijection that
// sends neighbouring ids far apart.
static uint32_t coprime_multiplier(uint32_t n)
{
    if (n < 4) return 1u;
    uint64_t m = (uint64_t)((double)n * 0.6180339887498949);
    if (m < 1) m = 1;
    for (;; ++m) {
        uint64_t a = m, b = n;
        while (b) {
            const uint64_t t = a % b;
            a = b;
            b = t;
        }
        if (a == 1) return (uint32_t)(m % n);
    }
}

// Does a sample k in [k0, k0 + cnt) of the axis have 0 < |value| < 2^-900 (binary32: 2^-100 after the cast)?  Round 1-4
// looked at every sample (4 096 per DataChunk, twice per submitted tile: 10 us of host time where an all-exterior tile costs
// the GPU 25).  The samples before the pinned last one are a monotonic sequence -- k * step is monotonic in k, and so are the
// roundings, the addition of `start` and the cast to float -- so the values nearest to zero sit at the sign change: two
// binary searches find the last sample below zero and the first above it.  (tests: mbk_view_needs_literal_doubling against
// the scan of every sample.)
static bool axis_has_tiny_nonzero(const Axis &a, uint32_t k0, uint32_t cnt, bool f32)
{
    const double thr = f32 ? 0x1p-100 : kSafeImagMin;
    auto value = [&](uint32_t k) { const double y = axis_value_host(a, k); return f32 ? (double)(float)y : y; };
    auto tiny = [&](double y) { return y != 0.0 && std::fabs(y) < thr; };
    if (cnt == 0) return false;
    uint32_t end = k0 + cnt;                    // (k0 + cnt <= a.n: the caller checked the window)
    if (a.n >= 1u && end == a.n) {              // the pinned last sample is not part of the monotonic sequence
        if (tiny(value(a.n - 1u))) return true;
        if (--end == k0) return false;
    }
    const double first = value(k0), last = value(end - 1u);
    if (tiny(first) || tiny(last)) return true;
    if ((first > 0.0 && last > 0.0) || (first < 0.0 && last < 0.0) || first == last) return false;   // no sign change inside
    const double s = last > first ? 1.0 : -1.0;   // s * value(k) is non-decreasing
    auto lower = [&](bool strict) {               // first k in [k0, end) with s * value(k) >= 0 (strict: > 0); end if none
        uint32_t lo = k0, hi = end;
        while (lo < hi) {
            const uint32_t mid = lo + (hi - lo) / 2u;
            const double y = s * value(mid);
            if (strict ? y > 0.0 : y >= 0.0) hi = mid;
            else lo = mid + 1u;
        }
        return lo;
    };
    const uint32_t ge = lower(false), gt = lower(true);
    if (ge > k0 && tiny(value(ge - 1u))) return true;     // the last sample on the far side of zero
    if (gt < end && tiny(value(gt))) return true;         // the first one on the near side
    return false;
}

static int validate_view(mbk_ctx *ctx, const mbk_view *v, bool *safe_doubling, bool f32 = false)
{
    if (!v) return fail(ctx, MBK_ERR_INVALID, "view is NULL");
    if (v->width == 0 || v->height == 0) return fail(ctx, MBK_ERR_INVALID, "empty view");
    if (v->ncols == 0 || v->nrows == 0) return fail(ctx, MBK_ERR_INVALID, "empty window");
    if ((uint64_t)v->col0 + v->ncols > v->width || (uint64_t)v->row0 + v->nrows > v->height)
        return fail(ctx, MBK_ERR_INVALID, "window exceeds the view");
    if ((uint64_t)v->ncols * v->nrows > (1ull << 31))
        return fail(ctx, MBK_ERR_INVALID, "window larger than 2^31 pixels");
    const double vals[6] = {v->start_r, v->start_i, v->range_r, v->range_i,
                            v->start_r + v->range_r, v->start_i + v->range_i};
    const double max_coord = f32 ? 0x1p60 : kMaxCoord;
    for (double x : vals)
        if (!std::isfinite(x) || std::fabs(x) > max_coord)
            return fail(ctx, MBK_ERR_INVALID, f32 ? "view coordinates must be finite and |x| <= 2^60 (fp32)"
                                                  : "view coordinates must be finite and |x| <= 2^500");
    *safe_doubling = axis_has_tiny_nonzero(make_axis(v->start_i, v->range_i, v->height), v->row0, v->nrows, f32);
    return MBK_OK;
}

static void free_scratch(StreamScratch &sc)
{
    if (sc.aux) (void)hipStreamSynchronize(sc.aux);
    for (int k = 0; k < kOrderRing; ++k) {
        if (sc.d_order[k]) (void)hipFree(sc.d_order[k]);
        if (sc.ev_cls[k]) (void)hipEventDestroy(sc.ev_cls[k]);
        if (sc.ev_done[k]) (void)hipEventDestroy(sc.ev_done[k]);
    }
    if (sc.aux) (void)hipStreamDestroy(sc.aux);
    if (sc.d_ctl) (void)hipFree(sc.d_ctl);
    if (sc.d_queues) (void)hipFree(sc.d_queues);
    if (sc.d_spill_z) (void)hipFree(sc.d_spill_z);
    if (sc.d_spill_meta) (void)hipFree(sc.d_spill_meta);
    if (sc.d_spill_cnt) (void)hipFree(sc.d_spill_cnt);
    if (sc.d_spill_chunks) (void)hipFree(sc.d_spill_chunks);
    if (sc.d_spill_src) (void)hipFree(sc.d_spill_src);
    if (sc.d_spill_total) (void)hipFree(sc.d_spill_total);
    if (sc.d_cursors) (void)hipFree(sc.d_cursors);
    if (sc.d_entries) (void)hipFree(sc.d_entries);
    if (sc.h_hint) (void)hipHostFree(sc.h_hint);
    if (sc.h_stamps) (void)hipHostFree(sc.h_stamps);
    if (sc.d_red) (void)hipFree(sc.d_red);
    if (sc.h_red) (void)hipHostFree(sc.h_red);
    sc = StreamScratch();
}

// The scratch set of `stream` (created on first use).  A caller that keeps creating streams would grow
// the table without bound, so past kMaxStreamScratch entries everything is drained and dropped.
static int get_scratch(mbk_ctx *ctx, hipStream_t stream, StreamScratch **out)
{
    for (StreamScratch &sc : ctx->scratch)
        if (sc.stream == stream) {
            *out = &sc;
            return MBK_OK;
        }
    if (ctx->scratch.size() >= kMaxStreamScratch) {
        MBK_HIP(ctx, hipDeviceSynchronize());
        for (StreamScratch &sc : ctx->scratch) free_scratch(sc);
        ctx->scratch.clear();
    }
    ctx->scratch.emplace_back();
    StreamScratch &sc = ctx->scratch.back();
    sc.stream = stream;
    *out = &sc;
    MBK_HIP(ctx, hipHostMalloc((void **)&sc.h_hint, 2 * sizeof(uint32_t), hipHostMallocDefault));
    sc.h_hint[0] = sc.h_hint[1] = 0xffffffffu;   // nothing known yet
    return MBK_OK;
}

// Window-dependent launch facts the kernels rely on (computed for the window a launch ACTUALLY covers: the
// edge strips of launch_refill and the bands of a view are windows of their own):
//   fast_bx_end / fast_by_end   leading whole 8x8 blocks that hold no pinned last sample (TileArgs comment);
//   ring_possible               conservative rectangle test against | |c|^2 - 4 | < margin (kernel margins are
//                               1e-9 fp64 / 1e-3 fp32 per pixel; the host test allows 1e-6 / 2e-3).
static void set_window_facts(TileArgs &a, bool f32);
static void shares_from_fractions(const double *f, uint32_t *cum);
static double window_heavy_share(const TileArgs &a);
static double window_heavy_share(mbk_ctx *ctx, const TileArgs &a);

// Kernel "units": new H fractions for the eight XCDs from the time stamps of the launches on this stream that have finished
// since the last look (mbk_units.h).  A launch that dealt XCD x the fraction f_x of the H list and saw it deal its last
// ids T_x after the launch's first workgroup started measures its speed as f_x / T_x; the fractions follow the normalised
// speeds with a gain of 0.3, clamped to +-12 % of an even deal.  A slot whose 72 stamps do not all carry the launch's
// number (still running, overwritten, a grid too small to reach the tail on the first trip) is skipped -- and so is every
// launch that did not have the chip to itself as far as this ctx can tell: with several streams in flight an XCD's stamps
// say when it found room for this launch among the others, not how fast it is (measured: the fractions then drift the
// wrong way, profiles/r04/xcd_balance_ab.txt), so a launch counts only if no tile launch of the ctx went to another stream
// between the units launch before it and the one after it.
static void xcd_shares_update(const mbk_ctx *ctx, StreamScratch &sc)
{
    const unsigned long long kMask = 0xffffffffffffull;
    uint32_t first = sc.xcd_issued >= kStampSlots ? sc.xcd_issued - kStampSlots + 1u : 1u;
    first = std::max(first, sc.xcd_consumed + 1u);
    for (uint32_t c = first; c <= sc.xcd_issued; ++c) {
        const volatile unsigned long long *st = sc.h_stamps + (size_t)(c % kStampSlots) * mbk::kStampWords;
        const StreamScratch::SharesUsed &used = sc.xcd_ring[c % kShareRing];
        if (used.seq != c) continue;
        const StreamScratch::SharesUsed &before = sc.xcd_ring[(c - 1u) % kShareRing], &after = sc.xcd_ring[(c + 1u) % kShareRing];
        const bool alone = c > 1u && before.seq == c - 1u && before.switches == used.switches &&
                           (c == sc.xcd_issued ? ctx->stream_switches == used.switches : after.seq == c + 1u && after.switches == used.switches);
        unsigned long long v[mbk::kStampWords];
        bool whole = true;
        for (uint32_t i = 0; i < mbk::kStampWords; ++i) {
            v[i] = st[i];
            whole = whole && (v[i] >> 48) == (c & 0xffffu);
        }
        if (!whole) continue;
        unsigned long long start = v[0] & kMask;
        for (uint32_t x = 1; x < 8u; ++x)
            if ((((v[x] & kMask) - start) & kMask) >> 47) start = v[x] & kMask;   // earlier (mod 2^48)
        double T[8], tmin = 1e300, tmax = 0.0, speed[8], sum = 0.0;
        for (uint32_t x = 0; x < 8u; ++x) {
            unsigned long long last = 0;
            for (uint32_t t = 0; t < mbk::kStampTail; ++t)
                last = std::max(last, ((v[8u + x * mbk::kStampTail + t] & kMask) - start) & kMask);
            T[x] = (double)last;
            tmin = std::min(tmin, T[x]);
            tmax = std::max(tmax, T[x]);
        }
        sc.xcd_consumed = c;
        if (!alone || tmin < 1000.0 || tmax > 1.25 * tmin) continue;   // under 10 us, or nothing a share could explain
        for (uint32_t x = 0; x < 8u; ++x) {
            speed[x] = (double)used.f[x] / T[x];
            sum += speed[x];
        }
        double fs = 0.0;
        for (uint32_t x = 0; x < 8u; ++x) {
            const double f = 0.7 * sc.xcd_f[x] + 0.3 * speed[x] / sum;
            sc.xcd_f[x] = std::min(0.14, std::max(0.11, f));
            fs += sc.xcd_f[x];
        }
        for (uint32_t x = 0; x < 8u; ++x) sc.xcd_f[x] /= fs;
    }
}

// Launch the one-wave-per-block kernels ("asm" / "group", fp64 or fp32) for the window described by `a`
// (a.col0/row0/ncols/nrows, output at a.out_*), optionally behind the heavy-first classify pre-pass.
// fuse / counts_unwanted / fused: as for launch_scan_t -- partial-result slots (already zeroed on `stream`) to which a kernel
// that can do so adds the tile's pixel-iterations and never-escaped count itself (the units kernel, when the int32 counts in
// `a` exist for the statistics only: it then writes none); *fused says whether it did.
static int launch_blocks(mbk_ctx *ctx, TileArgs a, uint32_t kernel, bool safe, bool f32, hipStream_t stream,
                         ReduceSlot *fuse = nullptr, bool counts_unwanted = false, bool *fused = nullptr)
{
    const uint32_t wpw = ctx->opt[MBK_OPT_WAVES_PER_WG];  // 8x8-pixel blocks (= waves) per workgroup
    a.blocks_x = (a.ncols + 8u * wpw - 1u) / (8u * wpw);
    const uint32_t by = (a.nrows + 7u) / 8u;
    set_window_facts(a, f32);
    const bool cyc = ctx->opt[MBK_OPT_CYCLE_DETECT] != 0u;
    const dim3 grid(a.blocks_x * by), block(64u * wpw);
    const uint32_t order_mode = ctx->opt[MBK_OPT_ORDER], probe_steps = ctx->opt[MBK_OPT_PROBE_STEPS];
    a.perm_mul = order_mode == 1 ? coprime_multiplier(grid.x) : 1u;
    a.order = nullptr;
    int order_slot = -1;
    bool mid_first = false, order_overlap = false;
    StreamScratch *order_sc = nullptr;
    // order 3 = "units" (mbk_units.h): the light blocks of eight neighbouring block columns are ONE workgroup.  Where the
    // units kernel cannot serve a launch (outputs, widths, step counts it has no form for) the launch takes order 2.
    bool units = order_mode == 3 && wpw == 1u && kernel == MBK_KERNEL_GROUP && !safe && a.smooth == nullptr &&
                 (a.counts || a.bytes) && !(a.bytes && a.quant_wide) && a.blocks_x % 8u == 0u && a.blocks_x <= 2048u &&
                 a.fast_bx_end > 0u && a.fast_by_end > 0u && (f32 || ctx->opt[MBK_OPT_GROUP_STEPS] == 16u) &&
                 grid.x >= 16384u && (uint32_t)a.mrd > 2u * probe_steps && by <= 0xffffu;
    // ... and only where there is light area to batch: the share of the host's 16 x 16 probe pixels of the window that is gone
    // after 4 steps (a deterministic function of the window).  Measured (profiles/r04/units_ab.txt): cfg2 (0.64 light) strict
    // -0.3 % / cycle test +4.7 %, DataChunk (1,0,0) (0.8) +2.1 % / +9.7 %, cfg3 (none) -0.7 % / 0.
    double unit_share = 0.0;
    if (units) {
        unit_share = window_heavy_share(ctx, a);
        units = (1.0 - unit_share) * 65536.0 >= (double)ctx->opt[MBK_OPT_UNITS_MIN_LIGHT];
    }
    if (order_mode >= 2 && grid.x >= 16384u && (uint32_t)a.mrd > 2u * probe_steps && a.blocks_x <= 0xffffu && by <= 0xffffu) {
        // (a list entry packs block row and workgroup column into 16 bits each; wider windows go in image order)
        // heavy-first dispatch order (see classify_blocks_kernel); small launches skip it: one kernel in image
        // order beats memset + classify + tile below ~16 k blocks (cfg1, 4096 blocks: 20 us against 27)
        StreamScratch *sc = nullptr;
        int rc = get_scratch(ctx, stream, &sc);
        if (rc != MBK_OK) return rc;
        // MBK_OPT_PREPASS_OVERLAP: 0 the pre-pass runs on the caller's stream, in order (one queue, no event: +4..5 us per cfg2
        // launch back to back, profiles/r06/prepass_modes_ab.txt); 1 / 2 on an auxiliary stream (2: at the highest priority), tied
        // to the tile kernel by events.  (Round 6 also tried it as an ANY-ORDER launch on the caller's stream -- an AQL packet
        // without the barrier bit, hipExtAnyOrderLaunch, which would start in the previous tile kernel's drain -- but gfx950
        // ignores the flag: profiles/microbench/anyorder.hip, profiles/r06/anyorder.txt.)
        const uint32_t pp_mode = ctx->opt[MBK_OPT_PREPASS_OVERLAP];
        const bool overlap = pp_mode != 0u;
        // MBK_OPT_PREPASS_OVERLAP = 2: the auxiliary stream has the highest priority, so that the pre-pass of launch L + 1 gets
        // its workgroups in while the tile kernel of launch L is in full swing instead of waiting for its drain (round 5: once
        // the launch no longer ends in a 20 us drain, a pre-pass that waited for it sits on the critical path)
        const uint32_t want_prio = pp_mode == 2u ? 2u : 1u;
        if (sc->pp_last != (int)pp_mode) {
            // the lists' hand-over differs by mode (events between two streams / the order of one queue): nothing of the old
            // mode may be in flight when the first launch of the new one picks its list
            if (sc->pp_last >= 0) {
                MBK_HIP(ctx, hipStreamSynchronize(stream));
                if (sc->aux) MBK_HIP(ctx, hipStreamSynchronize(sc->aux));
            }
            for (int k = 0; k < kOrderRing; ++k) sc->done_valid[k] = false;
            sc->pp_last = (int)pp_mode;
        }
        if (overlap && sc->aux && sc->aux_prio != want_prio) {
            MBK_HIP(ctx, hipStreamSynchronize(stream));
            MBK_HIP(ctx, hipStreamSynchronize(sc->aux));
            (void)hipStreamDestroy(sc->aux);
            sc->aux = nullptr;
            for (int k = 0; k < kOrderRing; ++k) {
                if (sc->ev_cls[k]) (void)hipEventDestroy(sc->ev_cls[k]);
                if (sc->ev_done[k]) (void)hipEventDestroy(sc->ev_done[k]);
                sc->ev_cls[k] = sc->ev_done[k] = nullptr;
                sc->done_valid[k] = false;
            }
        }
        if (overlap && !sc->aux) {
            if (want_prio == 2u) {
                int least = 0, greatest = 0;
                MBK_HIP(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
                MBK_HIP(ctx, hipStreamCreateWithPriority(&sc->aux, hipStreamNonBlocking, greatest));
            } else
            MBK_HIP(ctx, hipStreamCreateWithFlags(&sc->aux, hipStreamNonBlocking));
            sc->aux_prio = want_prio;
            for (int k = 0; k < kOrderRing; ++k) {
                MBK_HIP(ctx, hipEventCreateWithFlags(&sc->ev_cls[k], hipEventDisableTiming));
                MBK_HIP(ctx, hipEventCreateWithFlags(&sc->ev_done[k], hipEventDisableTiming));
            }
        }
        if (grid.x > sc->order_cap) {
            // the old lists may still be read / written by kernels in flight on the two streams
            MBK_HIP(ctx, hipStreamSynchronize(stream));
            if (sc->aux) MBK_HIP(ctx, hipStreamSynchronize(sc->aux));
            for (int k = 0; k < kOrderRing; ++k) {
                if (sc->d_order[k]) (void)hipFree(sc->d_order[k]);
                sc->d_order[k] = nullptr;
                sc->done_valid[k] = false;
            }
            sc->order_cap = 0;
            // list | 3 counters | middle-class list (classify_blocks_kernel) | 2 counters, the units kernel's plan (64-byte aligned)
            // | its settled H entries (mbk_units.h: units_settled_base)
            for (int k = 0; k < kOrderRing; ++k)
                MBK_HIP(ctx, hipMalloc((void **)&sc->d_order[k], mbk::units_list_words(grid.x) * sizeof(uint32_t)));
            sc->order_cap = grid.x;
        }
        // Pre-pass of THIS launch on the aux stream: it depends on the window only, not on anything the caller's
        // stream computes, so it may run while the previous launch's tile kernel is still busy (memset + classify
        // are 13 us of a 577 us cfg2 step, with the chip nearly idle).  List k is free once the tile kernel that
        // read it kOrderRing launches ago has finished (ev_done); the tile kernel waits for its list (ev_cls).  With
        // prepass_overlap = 0 everything goes to the caller's stream, as in rounds 1-2.
        const unsigned k = sc->order_turn++ % (unsigned)kOrderRing;
        uint32_t *ord = sc->d_order[k];
        uint32_t *cursors = ord + grid.x;   // right behind the list: the tile kernel finds them at order[gridDim.x]
        hipStream_t pre = overlap ? sc->aux : stream;
        if (overlap && sc->done_valid[k]) MBK_HIP(ctx, hipStreamWaitEvent(sc->aux, sc->ev_done[k], 0));
        // (serial mode needs no wait: the list's last reader, kOrderRing launches ago, ran on this same stream)
        if (units) {
            // MBK_OPT_M_LATE = s > 0: M blocks whose centre pixel escapes at step >= s open the dispatch order; MBK_OPT_H_SETTLED
            // = k > 0 (with the cycle test only): H blocks whose probe orbit is within 10^-k of settled close the front list
            const uint32_t hs = cyc ? ctx->opt[MBK_OPT_H_SETTLED] : 0u;
            const double settle_thr = hs ? std::pow(10.0, -(double)hs) : 0.0;
            // the pre-pass' six counters live apart from the lists (whose layout moves with the window's size), are zeroed once
            // here and put back to zero by the pre-pass itself (mbk_units.h: classify_units_kernel)
            if (!sc->d_ctl) {
                MBK_HIP(ctx, hipMalloc((void **)&sc->d_ctl, (size_t)kOrderRing * 8u * sizeof(uint32_t)));
                sc->ctl_dirty = true;
            }
            if (sc->ctl_dirty) {
                // ON the stream the pre-pass runs on (a hipMemset on the null stream does not order against a
                // non-blocking stream -- what commit d786d61 removed for the scan cursors), and behind whatever pre-pass may
                // still be running there and on the caller's stream; also after a failed launch (a pre-pass that did not run
                // to its last workgroup leaves its ticket behind, and no later one would ever clear it)
                if (sc->aux) MBK_HIP(ctx, hipStreamSynchronize(sc->aux));
                MBK_HIP(ctx, hipStreamSynchronize(stream));
                MBK_HIP(ctx, hipMemsetAsync(sc->d_ctl, 0, (size_t)kOrderRing * 8u * sizeof(uint32_t), pre));
                sc->ctl_dirty = false;
            }
            uint32_t *ctl = sc->d_ctl + 8u * k;
            // the shares of the eight XCDs: MBK_OPT_XCD_BALANCE 0 even, 1 following the stamps of earlier launches on this
            // stream, 2 a fixed uneven deal (tests)
            // (1 applies to the strict loops only: with the cycle test a launch ends with the drain of its boundary blocks, which
            // the stamps -- when an XCD dealt its last ids -- do not see; following them cost 0.3 % there, xcd_balance_ab.txt)
            // (and to a whole chip only: the stamps' "workgroup id mod 8 = XCD" is the dispatch order of 8 XCDs x 32 CUs -- a
            // partitioned device, CPX / DPX modes, deals differently: ADVICE r4)
            const uint32_t balance = ctx->opt[MBK_OPT_XCD_BALANCE] == 1u && (cyc || ctx->prop.multiProcessorCount != 256)
                                         ? 0u : ctx->opt[MBK_OPT_XCD_BALANCE];
            static const double kUneven[8] = {0.110, 0.140, 0.125, 0.120, 0.130, 0.125, 0.115, 0.135};
            if (balance == 1u && !sc->h_stamps) {
                MBK_HIP(ctx, hipHostMalloc((void **)&sc->h_stamps, (size_t)kStampSlots * mbk::kStampWords * sizeof(unsigned long long),
                                           hipHostMallocDefault));
                std::memset(sc->h_stamps, 0xff, (size_t)kStampSlots * mbk::kStampWords * sizeof(unsigned long long));
            }
            if (balance == 1u) xcd_shares_update(ctx, *sc);
            mbk::XcdShares w;
            double f[8];
            const uint32_t seq = ++sc->xcd_issued;
            StreamScratch::SharesUsed &used = sc->xcd_ring[seq % kShareRing];
            used.seq = seq;
            used.switches = ctx->stream_switches;
            for (uint32_t x = 0; x < 8u; ++x) {
                f[x] = balance == 1u ? sc->xcd_f[x] : (balance == 2u ? kUneven[x] : 0.125);
                used.f[x] = (float)f[x];
            }
            shares_from_fractions(f, w.cum);
            a.plan = ord + ((2u * (size_t)grid.x + 5u + 15u) & ~(size_t)15u);   // (ends below units_settled_base: + 16 + 40 <= + 64)
            a.stamps = balance == 1u ? sc->h_stamps + (size_t)(seq % kStampSlots) * mbk::kStampWords : nullptr;
            a.stamp_tag = seq & 0xffffu;
            const uint32_t cwg = ctx->opt[MBK_OPT_CLASSIFY_WG];
            hipLaunchKernelGGL(mbk::classify_units_kernel, dim3((grid.x + cwg - 1u) / cwg), dim3(cwg), 0, pre, a, grid.x,
                               (int32_t)probe_steps, ord, ctl, (int32_t)ctx->opt[MBK_OPT_M_LATE], settle_thr, ctl + 3, (uint32_t *)a.plan, w,
                               a.stamps ? 1u : 0u);
        } else {
        MBK_HIP(ctx, hipMemsetAsync(cursors, 0, 3 * sizeof(uint32_t), pre));
        // the middle class: MBK_OPT_PROBE_MID's (dispatched between heavy and light) or, without one, MBK_OPT_M_LATE's
        // (order 3 only: boundary blocks whose centre escapes at step >= m_late, dispatched first)
        mid_first = order_mode == 3 && ctx->opt[MBK_OPT_PROBE_MID] > probe_steps && ctx->opt[MBK_OPT_M_LATE] >= 2u &&
                    ctx->opt[MBK_OPT_M_LATE] <= probe_steps;
        hipLaunchKernelGGL(mbk::classify_blocks_kernel, dim3((grid.x + 1023u) / 1024u), dim3(1024), 0, pre, a,
                           grid.x, 8u * wpw, (int32_t)probe_steps, (int32_t)(mid_first ? ctx->opt[MBK_OPT_M_LATE] : ctx->opt[MBK_OPT_PROBE_MID]),
                           ord, cursors);
        }
        if (overlap) {
            MBK_HIP(ctx, hipEventRecord(sc->ev_cls[k], sc->aux));
            MBK_HIP(ctx, hipStreamWaitEvent(stream, sc->ev_cls[k], 0));
        }
        order_slot = (int)k;
        order_overlap = overlap;
        order_sc = sc;
        a.order = ord;
        a.ngrid = grid.x;
        a.order_mid = mid_first ? 2u : (ctx->opt[MBK_OPT_PROBE_MID] <= probe_steps ? 1u : 0u);
    }
    // SPILL (round 6; mbk_kernels.h: block_pixel_spill, mbk_spill.h): the launches the units kernel does not serve -- deep zooms:
    // little light area, long orbits -- with single-wave workgroups, 16-step groups (fp64) and counts / bytes as outputs, when
    // the depth and the size are worth a second pass and a checkpoint fits (the per-step prologue + the first checkpoint + 64 steps).
    // The second pass lasts at least as long as ONE wave needs for the steps a never-escaping lane has left -- ~ mrd x 21 ns, a
    // lone wave issues one dependent instruction per 8 cycles: 0.21 ms at mrd 10 000, 1.05 ms at 50 000 -- whatever the size of the
    // launch, while what the first pass saves grows with the number of blocks: the two meet near 2^19 blocks (measured: cfg3,
    // 2^20 blocks, +4.7 %; a 16 384 x 1 024 band of cfg4, 2^18 blocks at mrd 50 000, -4 %: profiles/r06/spill_ab.txt), hence
    // MBK_OPT_SPILL_MIN_BLOCKS.  Slot numbers are 32 bits, a lane's step count 26.
    StreamScratch *spill_sc = nullptr;
    const uint32_t spill_first = ctx->opt[MBK_OPT_SPILL_FIRST], spill_lanes = ctx->opt[MBK_OPT_SPILL_LANES];
    const bool spill = !units && a.order != nullptr && order_mode >= 2u && kernel == MBK_KERNEL_GROUP && !safe && a.smooth == nullptr && wpw == 1u &&
                       (a.counts || a.bytes) && spill_first != 0u && (f32 || ctx->opt[MBK_OPT_GROUP_STEPS] == 16u) &&
                       (uint32_t)a.mrd >= ctx->opt[MBK_OPT_SPILL_MIN_MRD] && (uint32_t)a.mrd < (1u << 26) &&
                       ((uint64_t)grid.x >> ctx->opt[MBK_OPT_SPILL_MIN_BLOCKS]) != 0ull &&
                       (uint64_t)a.mrd > (uint64_t)ctx->opt[MBK_OPT_EXACT_STEPS] + spill_first + 66u &&
                       (uint64_t)grid.x * spill_lanes < (1ull << 32) && a.fast_bx_end > 0u && a.fast_by_end > 0u;
    if (spill) {
        StreamScratch *sc = order_sc;
        const size_t slots = (size_t)grid.x * spill_lanes, nchunks = (grid.x + mbk::kSpillChunk - 1u) / mbk::kSpillChunk;
        if (slots > sc->spill_cap_slots || grid.x > sc->spill_cap_blocks) {
            MBK_HIP(ctx, hipStreamSynchronize(stream));   // the old buffers may still be in use
            for (void **q : {&sc->d_spill_z, (void **)&sc->d_spill_meta, (void **)&sc->d_spill_cnt, (void **)&sc->d_spill_chunks,
                             (void **)&sc->d_spill_src, (void **)&sc->d_spill_total}) {
                if (*q) (void)hipFree(*q);
                *q = nullptr;
            }
            sc->spill_cap_slots = sc->spill_cap_blocks = 0;
            // 24 bytes per slot (403 MB for an 8192^2 window): a device short of memory runs the launch without the second pass
            // -- an optimisation must not turn into an error -- and the next launch that wants it tries again
            const bool got = hipMalloc(&sc->d_spill_z, slots * 16u) == hipSuccess &&          // (zr, zi) as two doubles; two floats use half of it
                             hipMalloc((void **)&sc->d_spill_meta, slots * sizeof(uint32_t)) == hipSuccess &&
                             hipMalloc((void **)&sc->d_spill_src, slots * sizeof(uint32_t)) == hipSuccess &&
                             hipMalloc((void **)&sc->d_spill_cnt, (size_t)grid.x * sizeof(uint32_t)) == hipSuccess &&
                             hipMalloc((void **)&sc->d_spill_chunks, (nchunks * mbk::kSpillLevels + 1u) * sizeof(uint32_t)) == hipSuccess &&
                             hipMalloc((void **)&sc->d_spill_total, sizeof(unsigned long long)) == hipSuccess;
            if (got) {
                sc->spill_cap_slots = slots;
                sc->spill_cap_blocks = grid.x;
            } else {
                (void)hipGetLastError();   // (the failed allocation's sticky error)
                for (void **q : {&sc->d_spill_z, (void **)&sc->d_spill_meta, (void **)&sc->d_spill_cnt, (void **)&sc->d_spill_chunks,
                                 (void **)&sc->d_spill_src, (void **)&sc->d_spill_total}) {
                    if (*q) (void)hipFree(*q);
                    *q = nullptr;
                }
            }
        }
        if (sc->spill_cap_slots >= slots && sc->spill_cap_blocks >= grid.x && sc->d_spill_z) {
            a.spill_first = spill_first;
            a.spill_lanes = spill_lanes;
            a.spill_win_shift = ctx->opt[MBK_OPT_SPILL_CYC_SHIFT];
            a.spill_z = sc->d_spill_z;
            a.spill_meta = sc->d_spill_meta;
            a.spill_cnt = sc->d_spill_cnt;
            MBK_HIP(ctx, hipMemsetAsync(sc->d_spill_cnt, 0, (size_t)grid.x * sizeof(uint32_t), stream));
            spill_sc = sc;
        }
    }
    // MBK_OPT_WAVE_LIMIT: unused dynamic LDS caps the resident waves per SIMD (single-wave workgroups only)
    const uint32_t lds = wpw == 1u ? ctx->wave_limit_lds[ctx->opt[MBK_OPT_WAVE_LIMIT] & 7u] : 0u;
    if (units && a.order) {
        // Grid: the host cannot know how many units the probe makes of this window, so it estimates them from its own
        // 16 x 16 probe (share of the pixels still inside after 4 steps ~ the H and M blocks; the rest, eight to a unit)
        // and the kernel strides: a deterministic function of the window, no hint from earlier launches.
        const double share = unit_share;
        const double est = (double)grid.x * (share + (1.0 - share) / 8.0);
        const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
        uint32_t g = (uint32_t)std::min<double>((double)grid.x, est * 1.15 + 2048.0);
        // a multiple of 8: id mod 8 = workgroup mod 8 = the XCD; at least 2^14: bounds a wave's trips (the kernel's packed statistics)
        g = (std::max(std::max(g, std::min(grid.x, cus * 64u)), 16384u) + 7u) & ~7u;
        a.unit_stride = g;
        uint32_t qtab = 0u;   // quantised bytes of counts 1..4, packed (the light path's table)
        if (a.bytes && a.mrd > 0)
            for (uint32_t c = 1; c <= 4u; ++c)
                qtab |= (uint32_t)(((uint64_t)c * 256u + (uint32_t)a.mrd - 1u) / (uint32_t)a.mrd & 0xffu) << (8u * (c - 1u));
        const bool stats = fuse != nullptr && counts_unwanted && a.bytes != nullptr;
        if (stats) {
            a.stats = fuse;
            a.counts = nullptr;
            if (fused) *fused = true;
        }
#define MBK_LAUNCH_UNITS(T, G, CYC)                                                                                         \
    do {                                                                                                                   \
        if (stats)                                                                                                         \
            hipLaunchKernelGGL((mbk::tile_units_kernel<T, G, CYC, false, true, true>), dim3(g), dim3(64), lds, stream, a, qtab); \
        else if (a.counts && a.bytes)                                                                                      \
            hipLaunchKernelGGL((mbk::tile_units_kernel<T, G, CYC, true, true>), dim3(g), dim3(64), lds, stream, a, qtab);   \
        else if (a.bytes)                                                                                                  \
            hipLaunchKernelGGL((mbk::tile_units_kernel<T, G, CYC, false, true>), dim3(g), dim3(64), lds, stream, a, qtab);  \
        else                                                                                                               \
            hipLaunchKernelGGL((mbk::tile_units_kernel<T, G, CYC, true, false>), dim3(g), dim3(64), lds, stream, a, qtab);  \
    } while (0)
        if (f32 && cyc) MBK_LAUNCH_UNITS(float, 8, true);
        else if (f32) MBK_LAUNCH_UNITS(float, 8, false);
        else if (cyc) MBK_LAUNCH_UNITS(double, 16, true);
        else MBK_LAUNCH_UNITS(double, 16, false);
#undef MBK_LAUNCH_UNITS
    } else
    if (spill_sc && f32 && cyc)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<float, true, 8, true, true>), grid, block, lds, stream, a);
    else if (spill_sc && f32)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<float, true, 8, false, true>), grid, block, lds, stream, a);
    else if (spill_sc && cyc)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<double, true, 16, true, true>), grid, block, lds, stream, a);
    else if (spill_sc)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<double, true, 16, false, true>), grid, block, lds, stream, a);
    else
    if (f32 && safe)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<float, false, 0>), grid, block, lds, stream, a);
    else if (f32 && kernel == MBK_KERNEL_ASM)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<float, true, 0>), grid, block, lds, stream, a);
    else if (f32 && cyc)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<float, true, 8, true>), grid, block, lds, stream, a);
    else if (f32)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<float, true, 8>), grid, block, lds, stream, a);
    else if (safe)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<double, false, 0>), grid, block, lds, stream, a);
    else if (kernel == MBK_KERNEL_ASM)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<double, true, 0>), grid, block, lds, stream, a);
    else if (ctx->opt[MBK_OPT_GROUP_STEPS] == 32 && !cyc)   // (with the cycle test 32 means 16: mbk_loops.inc)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<double, true, 32>), grid, block, lds, stream, a);
    else if (ctx->opt[MBK_OPT_GROUP_STEPS] >= 16 && cyc)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<double, true, 16, true>), grid, block, lds, stream, a);
    else if (ctx->opt[MBK_OPT_GROUP_STEPS] >= 16)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<double, true, 16>), grid, block, lds, stream, a);
    else if (ctx->opt[MBK_OPT_GROUP_STEPS] == 8 && cyc)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<double, true, 8, true>), grid, block, lds, stream, a);
    else if (ctx->opt[MBK_OPT_GROUP_STEPS] == 8)
        hipLaunchKernelGGL((mbk::tile_asm_kernel<double, true, 8>), grid, block, lds, stream, a);
    else
        hipLaunchKernelGGL((mbk::tile_asm_kernel<double, true, 4>), grid, block, lds, stream, a);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            if (order_sc) order_sc->ctl_dirty = true;   // a pre-pass may be stuck half-way: clear its counters before the next one
            return fail(ctx, MBK_ERR_HIP, std::string("tile launch: ") + hipGetErrorString(e));
        }
    }
    if (order_slot >= 0 && order_overlap) {   // list `order_slot` is busy until this tile kernel has finished
        MBK_HIP(ctx, hipEventRecord(order_sc->ev_done[order_slot], stream));
        order_sc->done_valid[order_slot] = true;
    }
    if (spill_sc) {
        // the second pass: compact the blocks' spilled lanes into one list (chunk sums, their scan, the expansion), then run them
        StreamScratch *sc = spill_sc;
        const uint32_t nchunks = (grid.x + mbk::kSpillChunk - 1u) / mbk::kSpillChunk;
        hipLaunchKernelGGL(mbk::spill_count_kernel, dim3(nchunks), dim3(mbk::kSpillChunk), 0, stream, sc->d_spill_cnt, grid.x, nchunks, sc->d_spill_chunks);
        hipLaunchKernelGGL(mbk::rle_scan_kernel, dim3(1), dim3(1024), 0, stream, sc->d_spill_chunks, nchunks * mbk::kSpillLevels, sc->d_spill_total);
        hipLaunchKernelGGL(mbk::spill_expand_kernel, dim3(nchunks), dim3(mbk::kSpillChunk), 0, stream, sc->d_spill_cnt, grid.x, nchunks,
                           sc->d_spill_chunks, spill_lanes, sc->d_spill_src);
        // grid: an upper bound (the number of lanes is known on the device only), strided
        const uint32_t nwg = (uint32_t)std::min<uint64_t>(((uint64_t)grid.x * spill_lanes + 63u) / 64u, 16384u);
        // (fewer resident waves per SIMD for this pass -- through unused LDS, as MBK_OPT_WAVE_LIMIT -- were tried: 2 / 3 per SIMD cost
        // 0.1 / 0.05 ms of its 0.64 ms; what halved it was the order of the list, latest checkpoint first: profiles/r06/spill_ab.txt)
        const uint32_t lds2 = 0u;
        if (f32 && cyc)
            hipLaunchKernelGGL((mbk::tile_spill_kernel<float, true>), dim3(nwg), dim3(64), lds2, stream, a, sc->d_spill_src, sc->d_spill_total, nwg);
        else if (f32)
            hipLaunchKernelGGL((mbk::tile_spill_kernel<float, false>), dim3(nwg), dim3(64), lds2, stream, a, sc->d_spill_src, sc->d_spill_total, nwg);
        else if (cyc)
            hipLaunchKernelGGL((mbk::tile_spill_kernel<double, true>), dim3(nwg), dim3(64), lds2, stream, a, sc->d_spill_src, sc->d_spill_total, nwg);
        else
            hipLaunchKernelGGL((mbk::tile_spill_kernel<double, false>), dim3(nwg), dim3(64), lds2, stream, a, sc->d_spill_src, sc->d_spill_total, nwg);
        MBK_HIP(ctx, hipGetLastError());
        ++ctx->spill_launches;
        ctx->last_spill_stream = stream;
    }
    return MBK_OK;
}

// Can any pixel of the window lie within the ring | |c|^2 - 4 | < 1e-6 ?  (conservative rectangle test)
static bool window_may_touch_ring(const TileArgs &a, double margin = 1e-6)
{
    const double x0 = axis_value_host(a.re, a.col0), x1 = axis_value_host(a.re, a.col0 + a.ncols - 1u);
    const double y0 = axis_value_host(a.im, a.row0), y1 = axis_value_host(a.im, a.row0 + a.nrows - 1u);
    const double xlo = std::fmin(x0, x1), xhi = std::fmax(x0, x1), ylo = std::fmin(y0, y1), yhi = std::fmax(y0, y1);
    const double dx = (xlo <= 0.0 && 0.0 <= xhi) ? 0.0 : std::fmin(std::fabs(xlo), std::fabs(xhi));
    const double dy = (ylo <= 0.0 && 0.0 <= yhi) ? 0.0 : std::fmin(std::fabs(ylo), std::fabs(yhi));
    const double fx = std::fmax(std::fabs(xlo), std::fabs(xhi)), fy = std::fmax(std::fabs(ylo), std::fabs(yhi));
    const double rmin2 = dx * dx + dy * dy, rmax2 = fx * fx + fy * fy;
    return !(rmax2 < 4.0 - margin || rmin2 > 4.0 + margin);
}

static void set_window_facts(TileArgs &a, bool f32)
{
    a.ring_possible = window_may_touch_ring(a, f32 ? 2e-3 : 1e-6) ? 1u : 0u;
    a.fast_bx_end = a.fast_by_end = 0u;
    if (a.re.step_is_zero || a.im.step_is_zero) return;
    // window columns / rows before the axis' last sample (col0 + ncols <= n: validate_view); the last sample itself
    // counts when the regular formula reproduces it (axis_end_is_regular)
    const uint64_t re_n = (uint64_t)a.re.n - (axis_end_is_regular(a.re) ? 0u : 1u), im_n = (uint64_t)a.im.n - (axis_end_is_regular(a.im) ? 0u : 1u);
    const uint64_t cols_ok = std::min<uint64_t>(a.ncols, re_n - std::min<uint64_t>(a.col0, re_n));
    const uint64_t rows_ok = std::min<uint64_t>(a.nrows, im_n - std::min<uint64_t>(a.row0, im_n));
    a.fast_bx_end = (uint32_t)(cols_ok / 8u);
    a.fast_by_end = (uint32_t)(rows_ok / 8u);
}

// Kernel "refill": persistent lane-refill kernel on the interior blocks + "group" on the edge strips.
// Falls back to "group" for everything the narrow persistent kernel excludes (see mbk_persist.h).
static int launch_refill(mbk_ctx *ctx, const TileArgs &a, bool safe, hipStream_t stream)
{
    const bool eligible = !safe && a.mrd >= 2 && a.counts != nullptr && !a.re.step_is_zero &&
                          !a.im.step_is_zero && !window_may_touch_ring(a, 1e-6);
    // interior = whole 8x8 blocks that do not contain the last sample of either axis
    const uint32_t cols_ok = a.re.n > 0 ? std::min<uint64_t>(a.ncols, (uint64_t)(a.re.n - 1u) - std::min<uint64_t>(a.col0, a.re.n - 1u)) : 0u;
    const uint32_t rows_ok = a.im.n > 0 ? std::min<uint64_t>(a.nrows, (uint64_t)(a.im.n - 1u) - std::min<uint64_t>(a.row0, a.im.n - 1u)) : 0u;
    const uint32_t icols = eligible ? (cols_ok / 8u) * 8u : 0u, irows = eligible ? (rows_ok / 8u) * 8u : 0u;
    if (icols == 0 || irows == 0) return launch_blocks(ctx, a, MBK_KERNEL_GROUP, safe, false, stream);

    mbk::PersistArgs q;
    std::memset(&q, 0, sizeof(q));
    q.re_start = a.re.start;
    q.re_step = a.re.step;
    q.im_start = a.im.start;
    q.im_step = a.im.step;
    q.col0 = a.col0;
    q.row0 = a.row0;
    q.pitch = a.out_pitch;
    q.bxn = icols / 8u;
    q.nblocks = q.bxn * (irows / 8u);
    q.total = (uint32_t)a.mrd - 1u;
    q.livemin = ctx->opt[MBK_OPT_RF_LIVEMIN];
    q.patience = ctx->opt[MBK_OPT_RF_PATIENCE];
    q.batch = ctx->opt[MBK_OPT_RF_BATCH];
    q.counts = a.counts;
    StreamScratch *sc = nullptr;
    int rcs = get_scratch(ctx, stream, &sc);
    if (rcs != MBK_OK) return rcs;
    if (!sc->d_queues) MBK_HIP(ctx, hipMalloc((void **)&sc->d_queues, sizeof(WorkQueues)));
    WorkQueues *wq = sc->d_queues;
    hipLaunchKernelGGL(mbk::init_queues_kernel, dim3(1), dim3(64), 0, stream, wq, q.nblocks);
    uint32_t waves = (uint32_t)ctx->prop.multiProcessorCount * 4u * ctx->opt[MBK_OPT_RF_WAVES];
    if (waves > q.nblocks) waves = q.nblocks;
    hipLaunchKernelGGL(mbk::tile_persist_kernel, dim3((waves + 3u) / 4u), dim3(256), 0, stream, q, wq);
    MBK_HIP(ctx, hipGetLastError());
    // edge strips through the ordinary kernel, written into the same output image
    TileArgs e = a;
    e.bytes = nullptr;
    if (icols < a.ncols) {  // right strip, all rows
        e.col0 = a.col0 + icols;
        e.ncols = a.ncols - icols;
        e.out_col0 = a.out_col0 + icols;
        int rc = launch_blocks(ctx, e, MBK_KERNEL_GROUP, safe, false, stream);
        if (rc != MBK_OK) return rc;
    }
    if (irows < a.nrows) {  // bottom strip, interior columns only
        e = a;
        e.bytes = nullptr;
        e.ncols = icols;
        e.row0 = a.row0 + irows;
        e.nrows = a.nrows - irows;
        e.out_row0 = a.out_row0 + irows;
        int rc = launch_blocks(ctx, e, MBK_KERNEL_GROUP, safe, false, stream);
        if (rc != MBK_OK) return rc;
    }
    if (a.bytes) {
        const uint64_t npx = (uint64_t)a.ncols * a.nrows;
        hipLaunchKernelGGL(mbk::quantise_kernel, dim3(2048), dim3(256), 0, stream, a.counts, a.bytes, npx, a.mrd,
                           a.quant_wide, a.quant_rcp);
        MBK_HIP(ctx, hipGetLastError());
    }
    return MBK_OK;
}

// Share of the window that the light pass of kernel "scan" would NOT finish: a 16 x 16 grid of its pixels is
// iterated on the host for the 4 steps pass 1 runs (~1 000 steps, a few microseconds -- a launch costs more), and the
// fraction still inside is returned.  A scheduling heuristic only (kernel choice of MBK_KERNEL_DEFAULT): plain
// host doubles, no claim of bit-exactness, results never depend on it.
static double window_heavy_share(const TileArgs &a)
{
    const uint32_t k = 16;
    uint32_t inside = 0;
    for (uint32_t j = 0; j < k; ++j) {
        const double ci = axis_value_host(a.im, a.row0 + (uint32_t)(((uint64_t)(2u * j + 1u) * a.nrows) / (2u * k)));
        for (uint32_t i = 0; i < k; ++i) {
            const double cr = axis_value_host(a.re, a.col0 + (uint32_t)(((uint64_t)(2u * i + 1u) * a.ncols) / (2u * k)));
            double zr = cr, zi = ci;
            bool in = true;
            for (int n = 0; n < 4 && in; ++n) {
                const double t = zr * zr - zi * zi + cr;
                zi = 2.0 * zr * zi + ci;
                zr = t;
                in = zr * zr + zi * zi < 4.0;
            }
            inside += in ? 1u : 0u;
        }
    }
    return (double)inside / (double)(k * k);
}

// ... evaluated once per window: the last window's answer is kept on the ctx
static double window_heavy_share(mbk_ctx *ctx, const TileArgs &a)
{
    mbk_ctx::ProbeKey k;
    std::memset(&k, 0, sizeof(k));
    k.re = a.re;
    k.im = a.im;
    k.col0 = a.col0;
    k.row0 = a.row0;
    k.ncols = a.ncols;
    k.nrows = a.nrows;
    if (ctx->probe_valid && std::memcmp(&k, &ctx->probe_key, sizeof(k)) == 0) return ctx->probe_share;
    ctx->probe_key = k;
    ctx->probe_share = window_heavy_share(a);
    ctx->probe_valid = true;
    return ctx->probe_share;
}

// Kernel "scan": a persistent light pass over every 8x8 block, then one workgroup per block it listed as
// unfinished (mbk_scan.h).  Launches the light pass cannot serve go to launch_blocks ("group").
template <typename T>
// fuse: partial-result slots (already zeroed on `stream`) to which the finish-in-place form adds the tile's
// pixel-iterations and never-escaped count itself; *fused says whether it did.  counts_unwanted: the int32 counts in
// `a` exist for the statistics only -- a launch that fuses them does not write them.
static int launch_scan_t(mbk_ctx *ctx, TileArgs a, bool safe, hipStream_t stream, double probe_share,
                         ReduceSlot *fuse = nullptr, bool counts_unwanted = false, bool *fused = nullptr)
{
    const bool f32 = sizeof(T) == 4;
    a.blocks_x = (a.ncols + 7u) / 8u;
    const uint32_t nby = (a.nrows + 7u) / 8u;
    const uint32_t nblocks = a.blocks_x * nby;  // <= 2^31 / 64 (validate_view)
    const uint32_t total_steps = a.mrd > 1 ? (uint32_t)a.mrd - 1u : 0u;
    // Pass 1 is a persistent grid: exactly as many single-wave workgroups as are resident at once (a second
    // round of a statically strided pass would double its time) -- what the occupancy query reports for the
    // kernel, capped by the scan_waves option -- rounded down to whole block rows.
    const uint32_t cus = (uint32_t)ctx->prop.multiProcessorCount;
    // Nothing of the window's probe grid outlives the light pass (probe_share == 0: an all-exterior tile): pass 1
    // finishes its own leftovers and there is no pass 2 (tile_light_kernel, kInline)
    const bool inline_todo = probe_share == 0.0 && ctx->opt[MBK_OPT_SCAN_INLINE] != 0u;
    const uint32_t per_cu = std::min<uint32_t>(4u * ctx->opt[MBK_OPT_SCAN_WAVES],
                                               (uint32_t)(inline_todo ? ctx->scan_occ_inline[f32 ? 1 : 0] : ctx->scan_occ[f32 ? 1 : 0][0]));
    const uint32_t wmax = cus * per_cu;
    if (safe || a.re.step_is_zero || a.im.step_is_zero || a.smooth != nullptr || (a.bytes && a.quant_wide) ||
        total_steps < 4u || !(a.counts || a.bytes) || a.blocks_x > wmax)
        return launch_blocks(ctx, a, MBK_KERNEL_GROUP, safe, f32, stream);

    mbk::ScanArgs s;
    std::memset(&s, 0, sizeof(s));
    // Row strips (mbk_scan.h): in the finish-in-place form a wave's region is 64 x 1 pixels instead of an 8x8 block, so that
    // every store instruction writes one contiguous piece of a row; the fields below then count strips and rows.  Not for
    // narrow windows (the ragged last strip of a row runs with idle lanes).
    const bool strip = inline_todo && ctx->opt[MBK_OPT_SCAN_STRIP] != 0u && a.ncols >= 512u;
    const uint32_t regions_x = strip ? (a.ncols + 63u) / 64u : a.blocks_x, regions_y = strip ? a.nrows : nby, region_h = strip ? 1u : 8u;
    s.strip = strip ? 1u : 0u;
    s.strips_x = regions_x;
    s.stride_by = std::min(wmax / regions_x, regions_y);
    // (pass 1 addresses a run of blocks through a 32-bit lane offset that advances by this much per block)
    if ((uint64_t)s.stride_by * region_h * a.out_pitch * 4u >= (1ull << 31)) return launch_blocks(ctx, a, MBK_KERNEL_GROUP, safe, f32, stream);
    const uint32_t w1 = s.stride_by * regions_x;
    s.nblocks = nblocks;
    // XCD-aware column order (profiles/microbench/light_path.hip: the 8x8 store pattern of a 4096^2 int32 tile
    // takes 25.6 us in image order -- each 128-byte line is written by four XCDs -- and 12.9 us with it)
    s.xcd_map = (!strip && ctx->opt[MBK_OPT_SCAN_XCD_MAP] != 0u && a.blocks_x % 32u == 0u) ? 1u : 0u;
    // column jump: ~5/16 of the width, a multiple of 32 columns when the XCD map is on (keeps its grouping).  It
    // spreads the unfinished blocks of a window over the lists; where nothing is expected to be unfinished (the
    // finish-in-place form) a wave keeps its column: the jumps cost the all-exterior tile 3 of its 23 us
    // (profiles/r03/light_path_ab.txt).
    // (row strips wrap nothing: the column jump below counts 8x8 block columns, so it stays off with them -- strip implies
    // inline_todo today; the second condition keeps that true should the two ever be decoupled)
    s.col_period = (inline_todo || strip) ? 0u : ctx->opt[MBK_OPT_SCAN_COL_PERIOD];
    if (s.xcd_map) s.col_jump = 32u * (((a.blocks_x / 32u) * 5u / 16u) | 1u);
    else s.col_jump = std::max(1u, a.blocks_x * 5u / 16u);
    if (s.col_jump >= a.blocks_x) s.col_jump = 0u, s.col_period = 0u;
    s.fast_bx_end = a.ncols / (strip ? 64u : 8u);
    // block rows (strips: rows) whose imaginary coordinates all come from the regular formula (the asm computes them that way)
    const uint32_t im_n = a.im.n - (axis_end_is_regular(a.im) ? 0u : 1u);
    s.fast_by_end = std::min(a.nrows / region_h, im_n > a.row0 ? (im_n - a.row0) / region_h : 0u);
    s.qtab = 0u;
    if (a.bytes && a.mrd > 0)
        for (uint32_t k = 1; k <= 4u; ++k)
            s.qtab |= (uint32_t)(((uint64_t)k * 256u + (uint32_t)a.mrd - 1u) / (uint32_t)a.mrd & 0xffu) << (8u * (k - 1u));
    s.long_groups = ctx->opt[MBK_OPT_GROUP_STEPS] >= 16u ? 1u : 0u;   // (the scan path has no 32-step form: 32 means 16)
    set_window_facts(a, f32);
#define MBK_LAUNCH_LIGHT(INL)                                                                                              \
    do {                                                                                                                   \
        if (a.counts && a.bytes)                                                                                           \
            hipLaunchKernelGGL((mbk::tile_light_kernel<T, true, true, INL>), dim3(w1), dim3(64), 0, stream, a, s);         \
        else if (a.bytes)                                                                                                  \
            hipLaunchKernelGGL((mbk::tile_light_kernel<T, false, true, INL>), dim3(w1), dim3(64), 0, stream, a, s);        \
        else                                                                                                               \
            hipLaunchKernelGGL((mbk::tile_light_kernel<T, true, false, INL>), dim3(w1), dim3(64), 0, stream, a, s);        \
    } while (0)
    if (inline_todo) {
        // one launch, no lists, no scratch: a block the light path cannot finish is finished where it is found
        if (fuse && (a.bytes || !counts_unwanted)) {
            a.stats = fuse;
            if (counts_unwanted) a.counts = nullptr;
#define MBK_LAUNCH_LIGHT_STATS(INL)                                                                                        \
    do {                                                                                                                   \
        if (a.counts && a.bytes)                                                                                           \
            hipLaunchKernelGGL((mbk::tile_light_kernel<T, true, true, INL, true>), dim3(w1), dim3(64), 0, stream, a, s);   \
        else if (a.bytes)                                                                                                  \
            hipLaunchKernelGGL((mbk::tile_light_kernel<T, false, true, INL, true>), dim3(w1), dim3(64), 0, stream, a, s);  \
        else                                                                                                               \
            hipLaunchKernelGGL((mbk::tile_light_kernel<T, true, false, INL, true>), dim3(w1), dim3(64), 0, stream, a, s);  \
    } while (0)
            if (ctx->opt[MBK_OPT_CYCLE_DETECT] != 0u) MBK_LAUNCH_LIGHT_STATS(2);
            else MBK_LAUNCH_LIGHT_STATS(1);
#undef MBK_LAUNCH_LIGHT_STATS
            MBK_HIP(ctx, hipGetLastError());
            if (fused) *fused = true;
            return MBK_OK;
        }
        if (ctx->opt[MBK_OPT_CYCLE_DETECT] != 0u) MBK_LAUNCH_LIGHT(2);
        else MBK_LAUNCH_LIGHT(1);
        MBK_HIP(ctx, hipGetLastError());
        return MBK_OK;
    }

    // every list is fed by the waves with its index mod 64; a wave lists at most its own blocks
    const uint32_t qcap = ((w1 + mbk::kScanQueues - 1u) / mbk::kScanQueues) * ((nby + s.stride_by - 1u) / s.stride_by);
    StreamScratch *sc = nullptr;
    int rc = get_scratch(ctx, stream, &sc);
    if (rc != MBK_OK) return rc;
    if (!sc->d_cursors || sc->cursors_dirty) {
        if (!sc->d_cursors) MBK_HIP(ctx, hipMalloc((void **)&sc->d_cursors, 2 * sizeof(mbk::ScanCursors)));
        // once per stream, ON that stream: hipMemset on the null stream does not order against a
        // non-blocking stream (the first launch on a new stream raced with it and lost listed blocks).
        // Also after a failed launch: pass 1 of launch L is what clears the set of launch L+1, so a launch that
        // failed after taking its sets would leave the next one appending behind stale tails.
        MBK_HIP(ctx, hipMemsetAsync(sc->d_cursors, 0, 2 * sizeof(mbk::ScanCursors), stream));
        sc->scan_turn = 0;
        sc->cursors_dirty = false;
    }
    const size_t need = (size_t)qcap * mbk::kScanQueues;
    if (need > sc->scan_cap_blocks) {
        if (sc->d_entries) {
            MBK_HIP(ctx, hipStreamSynchronize(stream));  // may still be in use
            (void)hipFree(sc->d_entries);
        }
        sc->d_entries = nullptr;
        sc->scan_cap_blocks = 0;
        MBK_HIP(ctx, hipMalloc((void **)&sc->d_entries, need * sizeof(uint32_t)));
        sc->scan_cap_blocks = need;
    }
    s.cur = sc->d_cursors + (sc->scan_turn & 1u);
    s.cur_next = sc->d_cursors + ((sc->scan_turn + 1u) & 1u);
    ++sc->scan_turn;
    s.entries = sc->d_entries;
    s.qcap = qcap;
    // pass 2: 64 lists x (hint = longest list of the previous launch on this stream + 25 %).  Never fewer
    // workgroups than fill the chip (when the tile has that many blocks): a hint from a light tile followed by
    // a heavy one would otherwise leave a few waves looping over whole lists.
    // When the host probe of THIS window (window_heavy_share) found no pixel that outlives the light pass and the
    // finish-in-place form is switched off, pass 2 is launched as one workgroup per CU (its stride loop keeps it correct
    // if the probe missed a thin feature) instead of the chip-filling floor.
    const uint32_t hint = sc->h_hint[0];
    if (probe_share == 0.0) {
        s.ranks2 = std::min(qcap, (cus + mbk::kScanQueues - 1u) / mbk::kScanQueues);
    } else {
        s.ranks2 = hint == 0xffffffffu ? qcap : (uint32_t)std::min<uint64_t>(qcap, (uint64_t)hint + hint / 4u + 2u);
        s.ranks2 = std::max(s.ranks2, std::min(qcap, (cus * 32u + mbk::kScanQueues - 1u) / mbk::kScanQueues));
    }
    if (s.ranks2 == 0u) s.ranks2 = 1u;
    s.hint_out = sc->h_hint;
    MBK_LAUNCH_LIGHT(0);
#undef MBK_LAUNCH_LIGHT
    const uint32_t lds2 = ctx->wave_limit_lds[ctx->opt[MBK_OPT_WAVE_LIMIT] & 7u];
    if (ctx->opt[MBK_OPT_CYCLE_DETECT] != 0u)
        hipLaunchKernelGGL((mbk::tile_todo_kernel<T, true>), dim3(mbk::kScanQueues * s.ranks2), dim3(64), lds2, stream, a, s);
    else
        hipLaunchKernelGGL((mbk::tile_todo_kernel<T, false>), dim3(mbk::kScanQueues * s.ranks2), dim3(64), lds2, stream, a, s);
    {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) {
            sc->cursors_dirty = true;
            return fail(ctx, MBK_ERR_HIP, std::string("scan launch: ") + hipGetErrorString(e));
        }
    }
    return MBK_OK;
}

static int launch_tile(mbk_ctx *ctx, const mbk_view *v, uint32_t mrd, uint32_t flags,
                       int32_t *d_counts, uint8_t *d_bytes, hipStream_t stream, double *d_smooth = nullptr,
                       ReduceSlot *fuse = nullptr, bool counts_unwanted = false, bool *fused = nullptr)
{
    if (fused) *fused = false;
    if (stream != ctx->last_tile_stream) {
        ctx->last_tile_stream = stream;
        ++ctx->stream_switches;
    }
    bool safe = false;
    const bool f32 = (flags & MBK_PRECISION_F32) != 0;
    int rc = validate_view(ctx, v, &safe, f32);
    if (rc != MBK_OK) return rc;
    if (mrd > 0x7fffffffu) return fail(ctx, MBK_ERR_INVALID, "mrd must fit int32 (calc_mb_value returns int32)");
    const bool wc = (flags & MBK_WANT_COUNTS) != 0, wb = (flags & MBK_WANT_BYTES) != 0;
    if (!wc && !wb && !d_smooth) return fail(ctx, MBK_ERR_INVALID, "flags select no output");
    if (wc && !d_counts) return fail(ctx, MBK_ERR_INVALID, "MBK_WANT_COUNTS with NULL counts pointer");
    if (wb && !d_bytes) return fail(ctx, MBK_ERR_INVALID, "MBK_WANT_BYTES with NULL bytes pointer");
    if (wb && mrd == 0) return fail(ctx, MBK_ERR_INVALID, "mrd == 0 has no quantised form (division by zero)");

    TileArgs a;
    std::memset(&a, 0, sizeof(a));
    a.re = make_axis(v->start_r, v->range_r, v->width);
    a.im = make_axis(v->start_i, v->range_i, v->height);
    a.col0 = v->col0;
    a.row0 = v->row0;
    a.ncols = v->ncols;
    a.nrows = v->nrows;
    a.out_pitch = v->ncols;
    a.out_col0 = 0;
    a.out_row0 = 0;
    a.mrd = (int32_t)mrd;
    a.quant_wide = (mrd >= (1u << 23)) ? 1u : 0u;
    a.quant_rcp = mrd ? 1.0 / (double)mrd : 0.0;
    a.perm_mul = 1u;
    a.exact_steps = ctx->opt[MBK_OPT_EXACT_STEPS];
    a.exact_steps_long = std::min(ctx->opt[MBK_OPT_EXACT_STEPS], ctx->opt[MBK_OPT_EXACT_LONG]);
    a.cyc_window = ctx->opt[MBK_OPT_CYCLE_WINDOW];
    a.order = nullptr;
    a.counts = wc ? d_counts : nullptr;
    a.bytes = wb ? d_bytes : nullptr;
    a.smooth = d_smooth;

    const uint32_t kernel = flags & MBK_KERNEL_MASK;
    if (d_smooth && (f32 || kernel == MBK_KERNEL_SIMPLE || kernel == MBK_KERNEL_REFILL))
        return fail(ctx, MBK_ERR_INVALID, "smooth colouring is implemented by the fp64 scan / asm / group kernels only");
    if (f32 && (kernel == MBK_KERNEL_SIMPLE || kernel == MBK_KERNEL_REFILL))
        return fail(ctx, MBK_ERR_INVALID, "MBK_PRECISION_F32 is implemented by the scan / asm / group kernels only");
    switch (kernel) {
        case MBK_KERNEL_DEFAULT:
        case MBK_KERNEL_SCAN: {
            // Default = a deterministic function of the window (round 3; rounds 1-2 followed the heavy share the
            // PREVIOUS launch on the stream had reported, which made a launch's time depend on history and sent every
            // tile after a change -- a real pyramid alternates -- to the wrong kernel).  window_heavy_share probes
            // 256 pixels of the window for the 4 steps the light pass runs.  With more than ~1 %
            // (MBK_OPT_HEAVY_SHARE) of them unfinished the tile is bound by the arithmetic of its heavy blocks and the
            // light ones ride along for free in "group" (cfg2: scan 597 us, group 568); below that "group" is bound
            // by its 0.27 ns per workgroup and its stores, and "scan" wins (all-exterior tile: 29 us against 71).
            // Launches under 16 k blocks take "group", which for them is a single kernel in image order (cfg1: 20
            // us, scan 24).  heavy_share = 0 forces "group", 65536 forces "scan".
            if (kernel == MBK_KERNEL_DEFAULT && (uint64_t)((a.ncols + 7u) / 8u) * ((a.nrows + 7u) / 8u) < 16384u)
                return launch_blocks(ctx, a, MBK_KERNEL_GROUP, safe, f32, stream);
            const double share = window_heavy_share(ctx, a);
            if (kernel == MBK_KERNEL_DEFAULT &&
                (share * 65536.0 > (double)ctx->opt[MBK_OPT_HEAVY_SHARE] || ctx->opt[MBK_OPT_HEAVY_SHARE] == 0u))
                return launch_blocks(ctx, a, MBK_KERNEL_GROUP, safe, f32, stream, fuse, counts_unwanted, fused);
            return f32 ? launch_scan_t<float>(ctx, a, safe, stream, share, fuse, counts_unwanted, fused)
                       : launch_scan_t<double>(ctx, a, safe, stream, share, fuse, counts_unwanted, fused);
        }
        case MBK_KERNEL_GROUP:
        case MBK_KERNEL_ASM:
            return launch_blocks(ctx, a, kernel, safe, f32, stream, fuse, counts_unwanted, fused);
        case MBK_KERNEL_REFILL:
            return launch_refill(ctx, a, safe, stream);
        case MBK_KERNEL_SIMPLE: {
            a.blocks_x = (v->ncols + 31u) / 32u;
            const uint32_t by = (v->nrows + 7u) / 8u;
            const dim3 grid(a.blocks_x * by), block(256);
            hipLaunchKernelGGL(mbk::tile_simple_kernel<false>, grid, block, 0, stream, a);
            break;
        }
        default:
            return fail(ctx, MBK_ERR_INVALID, "unknown MBK_KERNEL_* selector");
    }
    MBK_HIP(ctx, hipGetLastError());
    return MBK_OK;
}

static int ensure_buffers(mbk_ctx *ctx, Slot &sl, size_t px)
{
    if (px <= sl.cap_px) return MBK_OK;
    if (sl.d_counts) (void)hipFree(sl.d_counts);
    if (sl.d_bytes) (void)hipFree(sl.d_bytes);
    sl.d_counts = nullptr;
    sl.d_bytes = nullptr;
    sl.cap_px = 0;
    MBK_HIP(ctx, hipMalloc((void **)&sl.d_counts, px * sizeof(int32_t)));
    MBK_HIP(ctx, hipMalloc((void **)&sl.d_bytes, px));
    sl.cap_px = px;
    return MBK_OK;
}

// clear = false: the partial results were zeroed earlier on this stream and a tile kernel has already added its fused
// share (pixel-iterations, never-escaped count); this pass then reads the bytes only.
static int launch_reduce(mbk_ctx *ctx, ReduceSlot *d_red, ReduceSlot *h_red, const int32_t *d_counts, const uint8_t *d_bytes,
                         uint64_t n, uint32_t mrd, hipStream_t stream, bool clear = true)
{
    if (clear) MBK_HIP(ctx, hipMemsetAsync(d_red, 0, sizeof(ReduceSlot) * mbk::kReduceSlots, stream));
    if (!d_counts && !d_bytes) {   // everything was fused into the tile kernel
        MBK_HIP(ctx, hipMemcpyAsync(h_red, d_red, sizeof(ReduceSlot) * mbk::kReduceSlots, hipMemcpyDeviceToHost, stream));
        return MBK_OK;
    }
    const bool vec = n >= 1024u && ((uintptr_t)d_counts & 15u) == 0u && ((uintptr_t)d_bytes & 3u) == 0u;
    uint64_t blocks = ((vec ? n / 4u : n) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks == 0) blocks = 1;
    // four pixels per lane and trip; the scalar kernel serves unaligned sub-buffers and tiny inputs
    if (vec && d_counts && d_bytes)
        hipLaunchKernelGGL((mbk::reduce_vec_kernel<true, true>), dim3((uint32_t)blocks), dim3(256), 0, stream, d_counts,
                           d_bytes, n, mrd, d_red);
    else if (vec && d_counts)
        hipLaunchKernelGGL((mbk::reduce_vec_kernel<true, false>), dim3((uint32_t)blocks), dim3(256), 0, stream, d_counts,
                           d_bytes, n, mrd, d_red);
    else if (vec && d_bytes)
        hipLaunchKernelGGL((mbk::reduce_vec_kernel<false, true>), dim3((uint32_t)blocks), dim3(256), 0, stream, d_counts,
                           d_bytes, n, mrd, d_red);
    else
        hipLaunchKernelGGL(mbk::reduce_kernel, dim3((uint32_t)blocks), dim3(256), 0, stream, d_counts,
                           d_bytes, n, mrd, d_red);
    MBK_HIP(ctx, hipGetLastError());
    MBK_HIP(ctx, hipMemcpyAsync(h_red, d_red, sizeof(ReduceSlot) * mbk::kReduceSlots, hipMemcpyDeviceToHost, stream));
    return MBK_OK;
}
static int launch_reduce(mbk_ctx *ctx, Slot &sl, const int32_t *d_counts, const uint8_t *d_bytes, uint64_t n,
                         uint32_t mrd, hipStream_t stream, bool clear = true)
{
    return launch_reduce(ctx, sl.d_red, sl.h_red, d_counts, d_bytes, n, mrd, stream, clear);
}

// the partial results of a finished reduction (the stream has been synchronised), added up
static ReduceOut reduce_total(const ReduceSlot *h_red)
{
    ReduceOut t = {};
    for (uint32_t k = 0; k < mbk::kReduceSlots; ++k) {
        const ReduceOut &r = h_red[k].r;
        t.pixel_iterations += r.pixel_iterations;
        t.never_pixels += r.never_pixels;
        t.run_starts += r.run_starts;
        t.any_byte_not_zero |= r.any_byte_not_zero;
        t.any_byte_not_one |= r.any_byte_not_one;
    }
    return t;
}

static ReduceOut reduce_total(const Slot &sl) { return reduce_total(sl.h_red); }

static void fill_stats_from_reduce(const ReduceSlot *h_red, mbk_stats *s, bool have_bytes)
{
    const ReduceOut t = reduce_total(h_red);
    s->pixel_iterations = t.pixel_iterations;
    s->never_pixels = t.never_pixels;
    s->all_bytes_zero = have_bytes && t.any_byte_not_zero == 0 ? 1u : 0u;
    s->all_bytes_one = have_bytes && t.any_byte_not_one == 0 ? 1u : 0u;
    s->rle_runs = have_bytes ? t.run_starts : 0ull;
}
static void fill_stats_from_reduce(const Slot &sl, mbk_stats *s, bool have_bytes)
{
    fill_stats_from_reduce(sl.h_red, s, have_bytes);
}

// ------------------------------------- C ABI ---------------------------------------------------

extern "C" {

int mbk_abi_version(void) { return MBK_ABI_VERSION; }

int mbk_device_count(int *count)
{
    if (!count) return fail(nullptr, MBK_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, MBK_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return MBK_OK;
}

int mbk_create(int device, mbk_ctx **out)
{
    if (!out) return fail(nullptr, MBK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, MBK_ERR_NO_DEVICE,
                    std::string("no HIP device visible (") + hipGetErrorString(e) +
                        "); libmbk_hip has no CPU fallback");
    if (device < 0 || device >= n) return fail(nullptr, MBK_ERR_NO_DEVICE, "device index out of range");
    mbk_ctx *ctx = new (std::nothrow) mbk_ctx();
    if (!ctx) return fail(nullptr, MBK_ERR_NOMEM, "out of host memory");
    ctx->device = device;
    static const uint32_t kDefaults[MBK_OPT_COUNT_] = {
        /* ORDER */ 3u, /* WAVES_PER_WG */ 1u, /* GROUP_STEPS */ 16u, /* EXACT_STEPS */ 8u, /* PROBE_STEPS */ 32u,
        /* SCAN_WAVES */ 8u, /* SCAN_XCD_MAP */ 1u, /* SCAN_COL_PERIOD */ 4u, /* HEAVY_SHARE */ 655u,
        /* RF_LIVEMIN */ 48u, /* RF_PATIENCE */ 256u, /* RF_BATCH */ 1u, /* RF_WAVES */ 8u, /* CYCLE_DETECT */ 1u,
        /* PROBE_MID */ 65537u, /* PREPASS_OVERLAP */ 1u, /* EXACT_LONG */ 0u, /* SCAN_INLINE */ 1u, /* WAVE_LIMIT */ 0u,
        /* UNITS_MIN_LIGHT */ 32768u, /* XCD_BALANCE */ 0u, /* M_LATE */ 8u, /* H_SETTLED */ 6u, /* CLASSIFY_WG */ 1024u,
        /* SCAN_STRIP */ 1u, /* CYCLE_WINDOW */ 32u, /* SPILL_FIRST */ 256u, /* SPILL_LANES */ 16u, /* SPILL_MIN_MRD */ 2048u, /* SPILL_MIN_BLOCKS */ 19u, /* SPILL_CYC_SHIFT */ 5u};
    std::memcpy(ctx->opt, kDefaults, sizeof(kDefaults));
#define MBK_CREATE_HIP(call)                                                        \
    do {                                                                            \
        hipError_t e2_ = (call);                                                    \
        if (e2_ != hipSuccess) {                                                    \
            fail(nullptr, MBK_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e2_)); \
            mbk_destroy(ctx);                                                       \
            return MBK_ERR_HIP;                                                     \
        }                                                                           \
    } while (0)
    MBK_CREATE_HIP(hipSetDevice(device));
    MBK_CREATE_HIP(hipGetDeviceProperties(&ctx->prop, device));
    if (std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        fail(nullptr, MBK_ERR_NO_DEVICE,
             std::string("device is ") + ctx->prop.gcnArchName + ", this library is built for gfx950 only");
        mbk_destroy(ctx);
        return MBK_ERR_NO_DEVICE;
    }
    for (Slot &sl : ctx->s) {
        MBK_CREATE_HIP(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        MBK_CREATE_HIP(hipEventCreate(&sl.ev_k0));
        MBK_CREATE_HIP(hipEventCreate(&sl.ev_k1));
        MBK_CREATE_HIP(hipEventCreate(&sl.ev_c0));
        MBK_CREATE_HIP(hipEventCreate(&sl.ev_c1));
        MBK_CREATE_HIP(hipMalloc((void **)&sl.d_red, sizeof(ReduceSlot) * mbk::kReduceSlots));
        MBK_CREATE_HIP(hipHostMalloc((void **)&sl.h_red, sizeof(ReduceSlot) * mbk::kReduceSlots, hipHostMallocDefault));
    }
    {
        const void *fns[2][2] = {{(const void *)mbk::tile_light_kernel<double, true, true>, (const void *)mbk::tile_todo_kernel<double>},
                                 {(const void *)mbk::tile_light_kernel<float, true, true>, (const void *)mbk::tile_todo_kernel<float>}};
        for (int f = 0; f < 2; ++f)
            for (int k = 0; k < 2; ++k) {
                int n = 0;
                MBK_CREATE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fns[f][k], 64, 0));
                ctx->scan_occ[f][k] = n > 0 ? n : 1;
            }
        // the finish-in-place launch is one of several instantiations (outputs x cycle test x fused statistics) with
        // their own register counts: size the persistent grid for the least resident of them, or a launch whose
        // instantiation holds fewer waves than assumed would run a second dispatch round
        const void *fi[2][6] = {{(const void *)mbk::tile_light_kernel<double, true, true, 2>, (const void *)mbk::tile_light_kernel<double, true, true, 1>,
                                 (const void *)mbk::tile_light_kernel<double, true, true, 2, true>, (const void *)mbk::tile_light_kernel<double, false, true, 2, true>,
                                 (const void *)mbk::tile_light_kernel<double, true, false, 2, true>, (const void *)mbk::tile_light_kernel<double, true, false, 1>},
                                {(const void *)mbk::tile_light_kernel<float, true, true, 2>, (const void *)mbk::tile_light_kernel<float, true, true, 1>,
                                 (const void *)mbk::tile_light_kernel<float, true, true, 2, true>, (const void *)mbk::tile_light_kernel<float, false, true, 2, true>,
                                 (const void *)mbk::tile_light_kernel<float, true, false, 2, true>, (const void *)mbk::tile_light_kernel<float, true, false, 1>}};
        for (int f = 0; f < 2; ++f) {
            int least = 0;
            for (int k = 0; k < 6; ++k) {
                int n = 0;
                MBK_CREATE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fi[f][k], 64, 0));
                n = n > 0 ? n : 1;
                least = (k == 0 || n < least) ? n : least;
            }
            ctx->scan_occ_inline[f] = least;
        }
    }
    {
        // MBK_OPT_WAVE_LIMIT: the smallest dynamic-LDS size per workgroup at which the occupancy calculator admits no more
        // than 4 k single-wave workgroups per CU (binary search on the runtime's own answer, no LDS granule assumed)
        const void *fn = (const void *)mbk::tile_asm_kernel<double, true, 16, false>;
        for (int k = 1; k <= 7; ++k) {
            uint32_t lo = 0, hi = 65536;   // occ(lo) > 4k >= occ(hi)
            int n = 0;
            MBK_CREATE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 64, hi));
            if (n > 4 * k) {
                ctx->wave_limit_lds[k] = hi;
                continue;
            }
            while (hi - lo > 64) {
                const uint32_t mid = (lo + hi) / 2;
                MBK_CREATE_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 64, mid));
                if (n > 4 * k) lo = mid;
                else hi = mid;
            }
            ctx->wave_limit_lds[k] = hi;
        }
    }
#undef MBK_CREATE_HIP
    *out = ctx;
    return MBK_OK;
}

void mbk_destroy(mbk_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
    for (Slot &sl : ctx->s) {
        if (sl.stream) (void)hipStreamSynchronize(sl.stream);
        if (sl.d_counts) (void)hipFree(sl.d_counts);
        if (sl.d_bytes) (void)hipFree(sl.d_bytes);
        if (sl.d_red) (void)hipFree(sl.d_red);
        if (sl.h_red) (void)hipHostFree(sl.h_red);
        if (sl.ev_k0) (void)hipEventDestroy(sl.ev_k0);
        if (sl.ev_k1) (void)hipEventDestroy(sl.ev_k1);
        if (sl.ev_c0) (void)hipEventDestroy(sl.ev_c0);
        if (sl.ev_c1) (void)hipEventDestroy(sl.ev_c1);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    if (!ctx->scratch.empty()) (void)hipDeviceSynchronize();  // caller streams may still use the scratch
    for (StreamScratch &sc : ctx->scratch) free_scratch(sc);
    if (ctx->d_rle) (void)hipFree(ctx->d_rle);
    if (ctx->d_smooth) (void)hipFree(ctx->d_smooth);
    delete ctx;
}

const char *mbk_last_error(const mbk_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int mbk_get_device_info(mbk_ctx *ctx, mbk_device_info *info)
{
    if (!ctx || !info) return fail(ctx, MBK_ERR_INVALID, "NULL argument");
    std::memset(info, 0, sizeof(*info));
    std::snprintf(info->name, sizeof(info->name), "%s", ctx->prop.name);
    std::snprintf(info->arch, sizeof(info->arch), "%s", ctx->prop.gcnArchName);
    info->compute_units = ctx->prop.multiProcessorCount;
    info->clock_mhz = ctx->prop.clockRate / 1000;
    info->wavefront_size = ctx->prop.warpSize;
    info->total_mem = ctx->prop.totalGlobalMem;
    return MBK_OK;
}

int mbk_device_pci_bus_id(mbk_ctx *ctx, char *buf, int len)
{
    if (!ctx || !buf || len < 16) return fail(ctx, MBK_ERR_INVALID, "buffer of at least 16 bytes needed");
    MBK_HIP(ctx, hipDeviceGetPCIBusId(buf, len, ctx->device));
    return MBK_OK;
}

int mbk_host_alloc(mbk_ctx *ctx, uint64_t bytes, void **out)
{
    if (!ctx || !out || bytes == 0) return fail(ctx, MBK_ERR_INVALID, "bad argument");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    MBK_HIP(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
    return MBK_OK;
}

int mbk_host_free(mbk_ctx *ctx, void *ptr)
{
    if (!ctx) return fail(ctx, MBK_ERR_INVALID, "ctx is NULL");
    if (!ptr) return MBK_OK;
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    MBK_HIP(ctx, hipHostFree(ptr));
    return MBK_OK;
}

int mbk_datachunk_geometry(uint32_t level, uint32_t index_real, uint32_t index_imag,
                           double *start_r, double *start_i, double *range)
{
    if (!start_r || !start_i || !range) return fail(nullptr, MBK_ERR_INVALID, "NULL output pointer");
    if (level == 0) return fail(nullptr, MBK_ERR_INVALID, "level must be > 0 (DataChunk.cs:99-100)");
    if (index_real >= level || index_imag >= level)
        return fail(nullptr, MBK_ERR_INVALID, "chunk index must be < level (DataChunk.cs:102-106)");
    // WorkerCUDA.py:75-78: chunk_range = (MAX_AXIS - MIN_AXIS) / level; start = MIN_AXIS + chunk_range*index
    volatile double chunk_range = (2.0 - (-2.0)) / (double)level;
    volatile double pr = chunk_range * (double)index_real;
    volatile double pi = chunk_range * (double)index_imag;
    *range = chunk_range;
    *start_r = -2.0 + pr;
    *start_i = -2.0 + pi;
    return MBK_OK;
}

int mbk_view_launch(mbk_ctx *ctx, const mbk_view *view, uint32_t mrd, uint32_t flags,
                    int32_t *d_counts, uint8_t *d_bytes, void *hip_stream)
{
    if (!ctx) return fail(ctx, MBK_ERR_INVALID, "ctx is NULL");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    return launch_tile(ctx, view, mrd, flags, d_counts, d_bytes, (hipStream_t)hip_stream);
}

// The window's TileArgs as far as the host-side tests need them (axes + window)
static TileArgs view_window_args(const mbk_view *v)
{
    TileArgs a;
    std::memset(&a, 0, sizeof(a));
    a.re = make_axis(v->start_r, v->range_r, v->width);
    a.im = make_axis(v->start_i, v->range_i, v->height);
    a.col0 = v->col0;
    a.row0 = v->row0;
    a.ncols = v->ncols;
    a.nrows = v->nrows;
    return a;
}

// Does every sample of the window lie outside the circle |c| = 2, with room to spare: |c|^2 >= 4 (1 + m), m = 1e-8
// (fp64) / 1e-4 (fp32)?  The samples of an axis lie between its two end samples up to a few units in the last place
// (k * step + start is monotonic in k before rounding), which the margin covers 10^7 times over.
static bool view_outside_circle2(const mbk_view *v, bool f32)
{
    const TileArgs a = view_window_args(v);
    const double x0 = axis_value_host(a.re, a.col0), x1 = axis_value_host(a.re, a.col0 + a.ncols - 1u);
    const double y0 = axis_value_host(a.im, a.row0), y1 = axis_value_host(a.im, a.row0 + a.nrows - 1u);
    const double xlo = std::fmin(x0, x1), xhi = std::fmax(x0, x1), ylo = std::fmin(y0, y1), yhi = std::fmax(y0, y1);
    const double dx = (xlo <= 0.0 && 0.0 <= xhi) ? 0.0 : std::fmin(std::fabs(xlo), std::fabs(xhi));
    const double dy = (ylo <= 0.0 && 0.0 <= yhi) ? 0.0 : std::fmin(std::fabs(ylo), std::fabs(yhi));
    const double rmin2 = dx * dx + dy * dy;
    return rmin2 >= 4.0 * (1.0 + (f32 ? 1e-4 : 1e-8));
}

static double view_probe_share(mbk_ctx *ctx, const mbk_view *v) { return window_heavy_share(ctx, view_window_args(v)); }

// enqueue kernel + reduction + D2H of one tile on a slot's stream (no host synchronisation)
static int submit_view(mbk_ctx *ctx, Slot &sl, const mbk_view *view, uint32_t mrd, uint32_t flags,
                       int32_t *h_counts, uint8_t *h_bytes)
{
    if (sl.busy) return fail(ctx, MBK_ERR_INVALID, "slot still has a tile in flight: call mbk_wait first");
    const bool wc = (flags & MBK_WANT_COUNTS) != 0, wb = (flags & MBK_WANT_BYTES) != 0;
    if (wc && !h_counts) return fail(ctx, MBK_ERR_INVALID, "MBK_WANT_COUNTS with NULL counts pointer");
    if (wb && !h_bytes) return fail(ctx, MBK_ERR_INVALID, "MBK_WANT_BYTES with NULL bytes pointer");
    bool dummy;
    int rc = validate_view(ctx, view, &dummy, (flags & MBK_PRECISION_F32) != 0);
    if (rc != MBK_OK) return rc;
    const size_t px = (size_t)view->ncols * view->nrows;
    const bool lazy = wb && (flags & MBK_LAZY_UNIFORM) != 0;
    const bool f32 = (flags & MBK_PRECISION_F32) != 0;
    sl.immediate = false;
    if (lazy && !wc && mrd >= 256u && mrd <= 0x7fffffffu && view_outside_circle2(view, f32)) {
        // Every pixel of the window has |c| >= 2 (1 + 1e-8): the reference's first update z1 = c^2 + c has
        // |z1| = |c| |c + 1| >= |c| (|c| - 1) > 2, so calc_mb_value returns 1 for each of them (WorkerCUDA.py:54-63; the proof
        // with the rounding errors of the five operations: tests/test_oracle.py::test_outside_circle_escapes_at_step_one)
        // and the quantised byte is ceil(256 / mrd) = 1 for mrd >= 256: DataChunk.cs:87's "Immediate" chunk.  With
        // MBK_LAZY_UNIFORM such a tile is not copied anyway, so nothing is enqueued at all: the corners of [-2, 2]^2 outside
        // the inscribed circle, 1 - pi/4 = 21 % of the tiles of a deep pyramid level (32 of level 16's 256).
        std::memset(&sl.imm, 0, sizeof(sl.imm));
        sl.imm.pixel_iterations = (uint64_t)px;
        sl.imm.all_bytes_one = 1u;
        sl.imm.rle_runs = 1ull;
        sl.immediate = true;
        sl.lazy_h_bytes = nullptr;
        sl.busy = true;
        sl.with_bytes = true;
        if (&sl == &ctx->s[0]) ctx->last_px = 0;   // (no bytes on the device: mbk_serialize_last has nothing to read)
        return MBK_OK;
    }
    rc = ensure_buffers(ctx, sl, px);
    if (rc != MBK_OK) return rc;
    // counts are always produced on the device (they feed the stats reduction); only what the
    // caller asked for crosses PCIe.
    const uint32_t dev_flags = (flags & (MBK_KERNEL_MASK | MBK_PRECISION_F32)) | MBK_WANT_COUNTS | (wb ? MBK_WANT_BYTES : 0u);
    // The statistics: a pass over counts (+ bytes) after the tile kernel -- unless the kernel that serves this window can
    // add up pixel-iterations and never-escaped pixels itself (the finish-in-place light pass: all-exterior tiles, 3 in 4
    // of a pyramid level).  Then no int32 count is written unless the caller asked for counts, and the pass reads the
    // bytes only (all-0 / all-1 flags, run count): 16 MiB instead of 80 for a DataChunk.
    MBK_HIP(ctx, hipMemsetAsync(sl.d_red, 0, sizeof(ReduceSlot) * mbk::kReduceSlots, sl.stream));
    bool fused = false;
    MBK_HIP(ctx, hipEventRecord(sl.ev_k0, sl.stream));
    rc = launch_tile(ctx, view, mrd, dev_flags, sl.d_counts, sl.d_bytes, sl.stream, nullptr, sl.d_red, !wc, &fused);
    if (rc != MBK_OK) return rc;
    MBK_HIP(ctx, hipEventRecord(sl.ev_k1, sl.stream));
    rc = launch_reduce(ctx, sl, fused ? nullptr : sl.d_counts, wb ? sl.d_bytes : nullptr, px, mrd, sl.stream, false);
    if (rc != MBK_OK) return rc;
    MBK_HIP(ctx, hipEventRecord(sl.ev_c0, sl.stream));
    // MBK_LAZY_UNIFORM: whether the tile is uniform is known only when its statistics are -- deciding the copy then costs a
    // synchronisation, an enqueue and a second synchronisation per tile (round 4: 20 % of a pyramid level's wall time).  So
    // the copy is enqueued NOW unless the host's 16 x 16 probe of the window says "every pixel gone within 4 steps" (the
    // all-exterior tile: counts 1..4, byte 1 everywhere for mrd >= 1024); a uniform tile that was copied anyway (an all-interior one)
    // wasted 0.3 ms of the copy engine beside its own 2 ms kernel, a non-uniform one that was not takes the old path in
    // mbk_wait.  Either way the caller is told by the stats whether h_bytes matters.
    bool lazy_deferred = false;
    if (lazy) lazy_deferred = view_probe_share(ctx, view) == 0.0;
    if (wb && !lazy_deferred) MBK_HIP(ctx, hipMemcpyAsync(h_bytes, sl.d_bytes, px, hipMemcpyDeviceToHost, sl.stream));
    if (wc) MBK_HIP(ctx, hipMemcpyAsync(h_counts, sl.d_counts, px * sizeof(int32_t), hipMemcpyDeviceToHost, sl.stream));
    MBK_HIP(ctx, hipEventRecord(sl.ev_c1, sl.stream));
    sl.lazy_h_bytes = lazy_deferred ? h_bytes : nullptr;
    sl.lazy_px = px;
    sl.busy = true;
    sl.with_bytes = wb;
    if (&sl == &ctx->s[0]) ctx->last_px = wb ? px : 0;
    return MBK_OK;
}

static int wait_slot(mbk_ctx *ctx, Slot &sl, mbk_stats *stats)
{
    if (!sl.busy) return fail(ctx, MBK_ERR_INVALID, "nothing was submitted on this slot");
    if (sl.immediate) {   // decided on the host at submit: nothing is in flight
        sl.busy = false;
        sl.immediate = false;
        if (stats) *stats = sl.imm;
        return MBK_OK;
    }
    MBK_HIP(ctx, hipStreamSynchronize(sl.stream));
    sl.busy = false;
    if (sl.lazy_h_bytes) {   // MBK_LAZY_UNIFORM: copy the bytes only if the tile is not all-0 / all-1
        uint8_t *dst = sl.lazy_h_bytes;
        sl.lazy_h_bytes = nullptr;
        const ReduceOut t = reduce_total(sl);
        if (t.any_byte_not_zero != 0 && t.any_byte_not_one != 0) {
            MBK_HIP(ctx, hipEventRecord(sl.ev_c0, sl.stream));
            MBK_HIP(ctx, hipMemcpyAsync(dst, sl.d_bytes, sl.lazy_px, hipMemcpyDeviceToHost, sl.stream));
            MBK_HIP(ctx, hipEventRecord(sl.ev_c1, sl.stream));
            MBK_HIP(ctx, hipStreamSynchronize(sl.stream));
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        MBK_HIP(ctx, hipEventElapsedTime(&stats->kernel_ms, sl.ev_k0, sl.ev_k1));
        MBK_HIP(ctx, hipEventElapsedTime(&stats->d2h_ms, sl.ev_c0, sl.ev_c1));
        fill_stats_from_reduce(sl, stats, sl.with_bytes);
    }
    return MBK_OK;
}

static void datachunk_view(mbk_view *v, double sr, double si, double range)
{
    v->start_r = sr;
    v->start_i = si;
    v->range_r = range;
    v->range_i = range;
    v->width = v->height = MBK_CHUNK_DEFINITION;
    v->col0 = v->row0 = 0;
    v->ncols = v->nrows = MBK_CHUNK_DEFINITION;
}

int mbk_view_compute(mbk_ctx *ctx, const mbk_view *view, uint32_t mrd, uint32_t flags,
                     int32_t *h_counts, uint8_t *h_bytes, mbk_stats *stats)
{
    if (!ctx || !view) return fail(ctx, MBK_ERR_INVALID, "NULL argument");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    int rc = submit_view(ctx, ctx->s[0], view, mrd, flags, h_counts, h_bytes);
    if (rc != MBK_OK) return rc;
    return wait_slot(ctx, ctx->s[0], stats);
}

int mbk_datachunk_submit(mbk_ctx *ctx, int slot, uint32_t level, uint32_t mrd, uint32_t index_real,
                         uint32_t index_imag, uint8_t *h_bytes, int32_t *h_counts)
{
    return mbk_datachunk_submit_ex(ctx, slot, level, mrd, index_real, index_imag, h_bytes, h_counts, 0u);
}

int mbk_datachunk_submit_ex(mbk_ctx *ctx, int slot, uint32_t level, uint32_t mrd, uint32_t index_real,
                            uint32_t index_imag, uint8_t *h_bytes, int32_t *h_counts, uint32_t flags)
{
    if (!ctx) return fail(ctx, MBK_ERR_INVALID, "ctx is NULL");
    if (slot < 0 || slot >= MBK_SLOTS) return fail(ctx, MBK_ERR_INVALID, "slot out of range");
    if (!h_bytes) return fail(ctx, MBK_ERR_INVALID, "h_bytes is NULL");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    double sr, si, range;
    int rc = mbk_datachunk_geometry(level, index_real, index_imag, &sr, &si, &range);
    if (rc != MBK_OK) {
        ctx->err = g_err;
        return rc;
    }
    mbk_view v;
    datachunk_view(&v, sr, si, range);
    return submit_view(ctx, ctx->s[slot], &v, mrd,
                       MBK_WANT_BYTES | (h_counts ? MBK_WANT_COUNTS : 0u) | (flags & (MBK_LAZY_UNIFORM | MBK_KERNEL_MASK)),
                       h_counts, h_bytes);
}

int mbk_view_submit(mbk_ctx *ctx, int slot, const mbk_view *view, uint32_t mrd, uint32_t flags,
                    int32_t *h_counts, uint8_t *h_bytes)
{
    if (!ctx || !view) return fail(ctx, MBK_ERR_INVALID, "NULL argument");
    if (slot < 0 || slot >= MBK_SLOTS) return fail(ctx, MBK_ERR_INVALID, "slot out of range");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    return submit_view(ctx, ctx->s[slot], view, mrd, flags, h_counts, h_bytes);
}

int mbk_view_needs_literal_doubling(const mbk_view *view, uint32_t flags, int *literal)
{
    if (!view || !literal) return fail(nullptr, MBK_ERR_INVALID, "NULL argument");
    bool safe = false;
    const int rc = validate_view(nullptr, view, &safe, (flags & MBK_PRECISION_F32) != 0);
    if (rc != MBK_OK) return rc;
    *literal = safe ? 1 : 0;
    return MBK_OK;
}

int mbk_view_outside_circle(const mbk_view *view, uint32_t flags, int *outside)
{
    if (!view || !outside) return fail(nullptr, MBK_ERR_INVALID, "NULL argument");
    bool dummy;
    const bool f32 = (flags & MBK_PRECISION_F32) != 0;
    const int rc = validate_view(nullptr, view, &dummy, f32);
    if (rc != MBK_OK) return rc;
    *outside = view_outside_circle2(view, f32) ? 1 : 0;
    return MBK_OK;
}

int mbk_wait(mbk_ctx *ctx, int slot, mbk_stats *stats)
{
    if (!ctx) return fail(ctx, MBK_ERR_INVALID, "ctx is NULL");
    if (slot < 0 || slot >= MBK_SLOTS) return fail(ctx, MBK_ERR_INVALID, "slot out of range");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    return wait_slot(ctx, ctx->s[slot], stats);
}

int mbk_datachunk(mbk_ctx *ctx, uint32_t level, uint32_t mrd, uint32_t index_real,
                  uint32_t index_imag, uint8_t *h_bytes, int32_t *h_counts, mbk_stats *stats)
{
    int rc = mbk_datachunk_submit(ctx, 0, level, mrd, index_real, index_imag, h_bytes, h_counts);
    if (rc != MBK_OK) return rc;
    return wait_slot(ctx, ctx->s[0], stats);
}

int mbk_view_launch_smooth(mbk_ctx *ctx, const mbk_view *view, uint32_t mrd, uint32_t flags,
                           int32_t *d_counts, double *d_smooth, void *hip_stream)
{
    if (!ctx || !d_smooth) return fail(ctx, MBK_ERR_INVALID, "NULL argument");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t f = (flags & (MBK_KERNEL_MASK | MBK_PRECISION_F32)) | (d_counts ? MBK_WANT_COUNTS : 0u);
    return launch_tile(ctx, view, mrd, f, d_counts, nullptr, (hipStream_t)hip_stream, d_smooth);
}

int mbk_view_compute_smooth(mbk_ctx *ctx, const mbk_view *view, uint32_t mrd, uint32_t flags,
                            int32_t *h_counts, double *h_smooth, mbk_stats *stats)
{
    if (!ctx || !view || !h_smooth) return fail(ctx, MBK_ERR_INVALID, "NULL argument");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    // the synchronous calls run on slot 0: its buffers, events and reduction scratch belong to a tile in flight
    if (ctx->s[0].busy) return fail(ctx, MBK_ERR_INVALID, "slot 0 has a tile in flight: call mbk_wait first");
    bool dummy;
    int rc = validate_view(ctx, view, &dummy);
    if (rc != MBK_OK) return rc;
    const size_t px = (size_t)view->ncols * view->nrows;
    rc = ensure_buffers(ctx, ctx->s[0], px);
    if (rc != MBK_OK) return rc;
    if (px > ctx->smooth_cap_px) {
        if (ctx->d_smooth) (void)hipFree(ctx->d_smooth);
        ctx->d_smooth = nullptr;
        ctx->smooth_cap_px = 0;
        MBK_HIP(ctx, hipMalloc((void **)&ctx->d_smooth, px * sizeof(double)));
        ctx->smooth_cap_px = px;
    }
    MBK_HIP(ctx, hipEventRecord(ctx->s[0].ev_k0, ctx->s[0].stream));
    rc = launch_tile(ctx, view, mrd, (flags & MBK_KERNEL_MASK) | MBK_WANT_COUNTS, ctx->s[0].d_counts, nullptr,
                     ctx->s[0].stream, ctx->d_smooth);
    if (rc != MBK_OK) return rc;
    MBK_HIP(ctx, hipEventRecord(ctx->s[0].ev_k1, ctx->s[0].stream));
    rc = launch_reduce(ctx, ctx->s[0], ctx->s[0].d_counts, nullptr, px, mrd, ctx->s[0].stream);
    if (rc != MBK_OK) return rc;
    MBK_HIP(ctx, hipEventRecord(ctx->s[0].ev_c0, ctx->s[0].stream));
    MBK_HIP(ctx, hipMemcpyAsync(h_smooth, ctx->d_smooth, px * sizeof(double), hipMemcpyDeviceToHost, ctx->s[0].stream));
    if (h_counts)
        MBK_HIP(ctx, hipMemcpyAsync(h_counts, ctx->s[0].d_counts, px * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->s[0].stream));
    MBK_HIP(ctx, hipEventRecord(ctx->s[0].ev_c1, ctx->s[0].stream));
    MBK_HIP(ctx, hipStreamSynchronize(ctx->s[0].stream));
    ctx->last_px = 0;
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        MBK_HIP(ctx, hipEventElapsedTime(&stats->kernel_ms, ctx->s[0].ev_k0, ctx->s[0].ev_k1));
        MBK_HIP(ctx, hipEventElapsedTime(&stats->d2h_ms, ctx->s[0].ev_c0, ctx->s[0].ev_c1));
        fill_stats_from_reduce(ctx->s[0], stats, false);
    }
    return MBK_OK;
}

int mbk_serialize_last(mbk_ctx *ctx, uint8_t *h_out, uint64_t cap, uint64_t *size, uint32_t *codec)
{
    if (!ctx || !h_out || !size || !codec) return fail(ctx, MBK_ERR_INVALID, "NULL argument");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->s[0].busy) return fail(ctx, MBK_ERR_INVALID, "slot 0 has a tile in flight: call mbk_wait first");
    const size_t n = ctx->last_px;
    if (n == 0) return fail(ctx, MBK_ERR_INVALID, "no tile with quantised bytes has been computed on this ctx");
    const uint32_t nblocks = (uint32_t)((n + mbk::kRleBlock - 1) / mbk::kRleBlock);
    // scratch layout (all 256-byte aligned): block counts | total | run_start | run_value | out
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // RLE is only emitted when 1 + 5*runs < 1 + n, i.e. for at most n/5 runs and n output bytes
    const size_t max_runs = n / 5 + 1;
    const size_t off_cnt = 0, off_tot = align(off_cnt + (size_t)nblocks * 4), off_start = align(off_tot + 8),
                 off_val = align(off_start + max_runs * 4), off_out = align(off_val + max_runs),
                 total_bytes = off_out + 1 + n;
    if (n > ctx->rle_cap_px) {
        if (ctx->d_rle) (void)hipFree(ctx->d_rle);
        ctx->d_rle = nullptr;
        ctx->rle_cap_px = 0;
        MBK_HIP(ctx, hipMalloc((void **)&ctx->d_rle, total_bytes));
        ctx->rle_cap_px = n;
    }
    uint32_t *d_cnt = (uint32_t *)(ctx->d_rle + off_cnt);
    unsigned long long *d_tot = (unsigned long long *)(ctx->d_rle + off_tot);
    uint32_t *d_start = (uint32_t *)(ctx->d_rle + off_start);
    uint8_t *d_val = ctx->d_rle + off_val, *d_out = ctx->d_rle + off_out;
    hipStream_t s = ctx->s[0].stream;
    hipLaunchKernelGGL(mbk::rle_count_kernel, dim3(nblocks), dim3(mbk::kRleBlock), 0, s, ctx->s[0].d_bytes, (uint64_t)n, d_cnt);
    hipLaunchKernelGGL(mbk::rle_scan_kernel, dim3(1), dim3(1024), 0, s, d_cnt, nblocks, d_tot);
    MBK_HIP(ctx, hipGetLastError());
    unsigned long long runs = 0;
    MBK_HIP(ctx, hipMemcpyAsync(&runs, d_tot, sizeof(runs), hipMemcpyDeviceToHost, s));
    MBK_HIP(ctx, hipStreamSynchronize(s));
    const uint64_t raw_size = 1 + (uint64_t)n, rle_size = 1 + 5 * (uint64_t)runs;
    // DataChunk.Serialize keeps the first serializer (Raw) unless a later one is strictly smaller
    const bool use_rle = rle_size < raw_size;
    *codec = use_rle ? MBK_CODEC_RLE : MBK_CODEC_RAW;
    *size = use_rle ? rle_size : raw_size;
    if (cap < *size) return fail(ctx, MBK_ERR_INVALID, "output buffer too small for the serialised chunk");
    if (use_rle) {
        hipLaunchKernelGGL(mbk::rle_scatter_kernel, dim3(nblocks), dim3(mbk::kRleBlock), 0, s, ctx->s[0].d_bytes,
                           (uint64_t)n, d_cnt, d_start, d_val);
        hipLaunchKernelGGL(mbk::rle_emit_kernel, dim3((uint32_t)((runs + 255) / 256)), dim3(256), 0, s, d_start,
                           d_val, (uint64_t)runs, (uint64_t)n, d_out);
        MBK_HIP(ctx, hipGetLastError());
        MBK_HIP(ctx, hipMemcpyAsync(h_out, d_out, rle_size, hipMemcpyDeviceToHost, s));
    } else {
        h_out[0] = MBK_CODEC_RAW;
        MBK_HIP(ctx, hipMemcpyAsync(h_out + 1, ctx->s[0].d_bytes, n, hipMemcpyDeviceToHost, s));
    }
    MBK_HIP(ctx, hipStreamSynchronize(s));
    return MBK_OK;
}

int mbk_set_option(mbk_ctx *ctx, int option, uint32_t value)
{
    if (!ctx) return fail(ctx, MBK_ERR_INVALID, "ctx is NULL");
    bool ok = false;
    switch (option) {
        case MBK_OPT_ORDER: ok = value <= 3u; break;
        case MBK_OPT_WAVES_PER_WG: ok = value == 1u || value == 2u || value == 4u; break;
        case MBK_OPT_GROUP_STEPS: ok = value == 4u || value == 8u || value == 16u || value == 32u; break;
        case MBK_OPT_EXACT_STEPS: ok = value <= 4096u; break;
        case MBK_OPT_PROBE_STEPS: ok = value >= 2u && value <= 65536u; break;
        case MBK_OPT_SCAN_WAVES: ok = value >= 1u && value <= 8u; break;
        case MBK_OPT_SCAN_XCD_MAP: ok = value <= 1u; break;
        case MBK_OPT_SCAN_COL_PERIOD: ok = value <= 65536u; break;
        case MBK_OPT_HEAVY_SHARE: ok = value <= 65536u; break;
        case MBK_OPT_RF_LIVEMIN: ok = value <= 63u; break;
        case MBK_OPT_RF_PATIENCE: ok = value >= 16u && value <= (1u << 20); break;
        case MBK_OPT_RF_BATCH: ok = value >= 1u && value <= 64u; break;
        case MBK_OPT_RF_WAVES: ok = value >= 1u && value <= 8u; break;
        case MBK_OPT_CYCLE_DETECT: ok = value <= 1u; break;
        case MBK_OPT_PROBE_MID: ok = value >= 2u && value <= 65537u; break;
        case MBK_OPT_PREPASS_OVERLAP: ok = value <= 2u; break;
        case MBK_OPT_CLASSIFY_WG: ok = value >= 64u && value <= 1024u && value % 64u == 0u; break;
        case MBK_OPT_SCAN_STRIP: ok = value <= 1u; break;
        case MBK_OPT_CYCLE_WINDOW: ok = value <= 65536u; break;
        case MBK_OPT_EXACT_LONG: ok = value <= 4096u; break;
        case MBK_OPT_SCAN_INLINE: ok = value <= 1u; break;
        case MBK_OPT_WAVE_LIMIT: ok = value <= 7u; break;
        case MBK_OPT_UNITS_MIN_LIGHT: ok = value <= 65536u; break;
        case MBK_OPT_XCD_BALANCE: ok = value <= 2u; break;
        case MBK_OPT_M_LATE: ok = value <= 65536u; break;
        case MBK_OPT_H_SETTLED: ok = value <= 30u; break;
        case MBK_OPT_SPILL_FIRST: ok = value <= 65536u && value % 32u == 0u; break;
        case MBK_OPT_SPILL_LANES: ok = value >= 1u && value <= 32u; break;
        case MBK_OPT_SPILL_MIN_MRD: ok = true; break;
        case MBK_OPT_SPILL_MIN_BLOCKS: ok = value <= 31u; break;
        case MBK_OPT_SPILL_CYC_SHIFT: ok = value <= 31u; break;
        default: return fail(ctx, MBK_ERR_INVALID, "unknown MBK_OPT_* selector");
    }
    if (!ok) return fail(ctx, MBK_ERR_INVALID, "option value out of range");
    ctx->opt[option] = value;
    return MBK_OK;
}

int mbk_get_option(mbk_ctx *ctx, int option, uint32_t *value)
{
    if (!ctx || !value) return fail(ctx, MBK_ERR_INVALID, "NULL argument");
    if (option >= MBK_INFO_SCAN_WG_PER_CU && option < MBK_INFO_SCAN_WG_PER_CU + 4) {
        const int k = option - MBK_INFO_SCAN_WG_PER_CU;   // [f64 scan, f64 heavy, f32 scan, f32 heavy]
        *value = (uint32_t)ctx->scan_occ[k >> 1][k & 1];
        return MBK_OK;
    }
    if (option >= MBK_INFO_XCD_SHARE && option < MBK_INFO_XCD_SHARE + 10) {
        // the stream whose units launches have left the most time stamps so far
        const StreamScratch *best = nullptr;
        for (const StreamScratch &sc : ctx->scratch)
            if (!best || sc.xcd_consumed > best->xcd_consumed) best = &sc;
        const int k = option - MBK_INFO_XCD_SHARE;
        *value = !best ? 0u : k < 8 ? (uint32_t)std::lround(best->xcd_f[k] * 1048576.0) : k == 8 ? best->xcd_consumed : best->xcd_issued;
        return MBK_OK;
    }
    if (option == MBK_INFO_SPILL || option == MBK_INFO_SPILL + 1) {
        *value = ctx->spill_launches;
        if (option == MBK_INFO_SPILL) {   // lanes the last such launch handed to its second pass (waits for that launch)
            *value = 0u;
            for (const StreamScratch &sc : ctx->scratch)
                if (sc.stream == ctx->last_spill_stream && sc.d_spill_total && ctx->spill_launches) {
                    unsigned long long t = 0;
                    MBK_HIP(ctx, hipSetDevice(ctx->device));
                    MBK_HIP(ctx, hipStreamSynchronize(sc.stream));
                    MBK_HIP(ctx, hipMemcpy(&t, sc.d_spill_total, sizeof(t), hipMemcpyDeviceToHost));
                    *value = (uint32_t)std::min<unsigned long long>(t, 0xffffffffull);
                }
        }
        return MBK_OK;
    }
    if (option < 0 || option >= MBK_OPT_COUNT_) return fail(ctx, MBK_ERR_INVALID, "unknown MBK_OPT_* selector");
    *value = ctx->opt[option];
    return MBK_OK;
}

// The share arithmetic of the units kernel on the host (no device, no context): the very functions classify_units_kernel and
// tile_units_kernel call, for the CPU tests.
static void shares_from_fractions(const double *f, uint32_t *cum)
{
    double c = 0.0;
    for (uint32_t x = 0; x < 8u; ++x) {
        c += f[x];
        cum[x] = x == 7u ? (1u << 24) : (uint32_t)std::lround(std::min(1.0, std::max(0.0, c)) * (double)(1u << 24));
    }
}

int mbk_units_plan(uint32_t n_h, uint32_t n_v, uint32_t n_m, const double *fractions, uint32_t *plan)
{
    if (!fractions || !plan) return MBK_ERR_INVALID;
    if ((uint64_t)n_h + n_v + n_m > 0x7fffffffull) return MBK_ERR_INVALID;
    uint32_t cum[8];
    shares_from_fractions(fractions, cum);
    mbk::units_plan(n_h, n_v, n_m, cum, 0u, plan, 0u, 0u);
    return MBK_OK;
}

int mbk_units_lookup(const uint32_t *plan, uint32_t id, uint32_t *list, uint32_t *index)
{
    if (!plan || !list || !index) return MBK_ERR_INVALID;
    const uint32_t x = id & 7u;
    bool is_h = false;
    uint32_t i = 0;
    *list = 0u;
    *index = 0u;
    if (id >= plan[2] || !mbk::units_lookup(id, plan[3], plan[4], plan[8u + x], plan[16u + x], plan[24u + x], plan[32u + x], is_h, i))
        return MBK_OK;
    *list = is_h ? 1u : (i < plan[1] ? 2u : 3u);
    *index = is_h ? i : (i < plan[1] ? i : i - plan[1]);
    return MBK_OK;
}

int mbk_quantise_counts(mbk_ctx *ctx, const int32_t *h_counts, uint64_t n, uint32_t mrd, uint8_t *h_bytes)
{
    if (!ctx || !h_counts || !h_bytes) return fail(ctx, MBK_ERR_INVALID, "NULL argument");
    if (mrd == 0 || mrd > 0x7fffffffu) return fail(ctx, MBK_ERR_INVALID, "mrd must be in [1, 2^31)");
    if (n == 0) return MBK_OK;
    if (n > (1ull << 31)) return fail(ctx, MBK_ERR_INVALID, "more than 2^31 counts");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    Slot &sl = ctx->s[0];
    if (sl.busy) return fail(ctx, MBK_ERR_INVALID, "slot 0 has a tile in flight: call mbk_wait first");
    int rc = ensure_buffers(ctx, sl, (size_t)n);
    if (rc != MBK_OK) return rc;
    MBK_HIP(ctx, hipMemcpyAsync(sl.d_counts, h_counts, n * sizeof(int32_t), hipMemcpyHostToDevice, sl.stream));
    hipLaunchKernelGGL(mbk::quantise_kernel, dim3(2048), dim3(256), 0, sl.stream, sl.d_counts, sl.d_bytes, n,
                       (int32_t)mrd, mrd >= (1u << 23) ? 1u : 0u, 1.0 / (double)mrd);
    MBK_HIP(ctx, hipGetLastError());
    MBK_HIP(ctx, hipMemcpyAsync(h_bytes, sl.d_bytes, n, hipMemcpyDeviceToHost, sl.stream));
    MBK_HIP(ctx, hipStreamSynchronize(sl.stream));
    ctx->last_px = 0;
    return MBK_OK;
}

int mbk_reduce_counts(mbk_ctx *ctx, const int32_t *d_counts, uint64_t n, uint32_t mrd,
                      void *hip_stream, mbk_stats *stats)
{
    if (!ctx || !d_counts || !stats) return fail(ctx, MBK_ERR_INVALID, "NULL argument");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)hip_stream;
    // partial results live in the STREAM's scratch (not in slot 0's: a tile in flight on slot 0 reduces into those)
    StreamScratch *sc = nullptr;
    int rc = get_scratch(ctx, s, &sc);
    if (rc != MBK_OK) return rc;
    if (!sc->d_red) {
        MBK_HIP(ctx, hipMalloc((void **)&sc->d_red, sizeof(ReduceSlot) * mbk::kReduceSlots));
        MBK_HIP(ctx, hipHostMalloc((void **)&sc->h_red, sizeof(ReduceSlot) * mbk::kReduceSlots, hipHostMallocDefault));
    }
    rc = launch_reduce(ctx, sc->d_red, sc->h_red, d_counts, nullptr, n, mrd, s);
    if (rc != MBK_OK) return rc;
    MBK_HIP(ctx, hipStreamSynchronize(s));
    std::memset(stats, 0, sizeof(*stats));
    fill_stats_from_reduce(sc->h_red, stats, false);
    return MBK_OK;
}

// ---- the native worker loop (mbk_feeder.h) ----------------------------------------------------------------

static int ctx_submit(void *user, int slot, uint32_t level, uint32_t mrd, uint32_t ir, uint32_t ii, uint8_t *h_bytes)
{
    return mbk_datachunk_submit_ex((mbk_ctx *)user, slot, level, mrd, ir, ii, h_bytes, nullptr, MBK_LAZY_UNIFORM);
}
static int ctx_wait(void *user, int slot, mbk_stats *stats) { return mbk_wait((mbk_ctx *)user, slot, stats); }
static void *ctx_alloc(void *user, uint64_t bytes)
{
    void *p = nullptr;
    return mbk_host_alloc((mbk_ctx *)user, bytes, &p) == MBK_OK ? p : nullptr;
}
static void ctx_release(void *user, void *ptr) { (void)mbk_host_free((mbk_ctx *)user, ptr); }

int mbk_net_set_option(int option, uint32_t value)
{
    mbkf::NetConfig &c = mbkf::net();
    switch (option) {
        case MBK_NET_MAX_CONNECTIONS:
            if (value < 1u || value > 64u) return fail(nullptr, MBK_ERR_INVALID, "max connections must be in 1..64");
            c.max_connections.store(value);
            {
                std::lock_guard<std::mutex> g(c.m);
                c.peak = c.open;
            }
            c.cv.notify_all();
            return MBK_OK;
        case MBK_NET_CONNECT_TIMEOUT_MS: c.connect_timeout_ms.store(value); return MBK_OK;
        case MBK_NET_IO_TIMEOUT_MS: c.io_timeout_ms.store(value); return MBK_OK;
        case MBK_NET_RETRIES:
            if (value > 100u) return fail(nullptr, MBK_ERR_INVALID, "retries must be in 0..100");
            c.retries.store(value);
            return MBK_OK;
        case MBK_NET_BACKOFF_MS:
            if (value < 1u || value > 10000u) return fail(nullptr, MBK_ERR_INVALID, "backoff must be in 1..10000 ms");
            c.backoff_ms.store(value);
            return MBK_OK;
        case MBK_NET_STOP: c.stop.store(value ? 1u : 0u); return MBK_OK;
        case MBK_NET_FEEDER_SLOTS:
            if (value < 1u || value > 8u) return fail(nullptr, MBK_ERR_INVALID, "feeder slots must be in 1..8");
            c.feeder_slots.store(value);
            return MBK_OK;
        default: return fail(nullptr, MBK_ERR_INVALID, "unknown MBK_NET_* selector");
    }
}

int mbk_net_get_option(int option, uint32_t *value)
{
    if (!value) return fail(nullptr, MBK_ERR_INVALID, "NULL argument");
    mbkf::NetConfig &c = mbkf::net();
    switch (option) {
        case MBK_NET_MAX_CONNECTIONS: *value = c.max_connections.load(); return MBK_OK;
        case MBK_NET_CONNECT_TIMEOUT_MS: *value = c.connect_timeout_ms.load(); return MBK_OK;
        case MBK_NET_IO_TIMEOUT_MS: *value = c.io_timeout_ms.load(); return MBK_OK;
        case MBK_NET_RETRIES: *value = c.retries.load(); return MBK_OK;
        case MBK_NET_BACKOFF_MS: *value = c.backoff_ms.load(); return MBK_OK;
        case MBK_NET_STOP: *value = c.stop.load(); return MBK_OK;
        case MBK_NET_FEEDER_SLOTS: *value = c.feeder_slots.load(); return MBK_OK;
        case MBK_NET_PEAK_CONNECTIONS: {
            std::lock_guard<std::mutex> g(c.m);
            *value = c.peak;
            return MBK_OK;
        }
        default: return fail(nullptr, MBK_ERR_INVALID, "unknown MBK_NET_* selector");
    }
}

int mbk_feeder_run(const mbk_feeder_ops *ops, const char *addr, uint16_t port, uint64_t max_tiles, uint32_t senders,
                   mbk_worker_report *report)
{
    if (!ops || !ops->submit || !ops->wait || !ops->alloc || !ops->release || !addr)
        return fail(nullptr, MBK_ERR_INVALID, "NULL argument");
    std::string err;
    const int rc = mbkf::run(ops, addr, port, max_tiles, senders, report, &err, mbkf::net().feeder_slots.load());
    if (rc != MBK_OK) return fail(nullptr, rc, err);
    return MBK_OK;
}

int mbk_worker_run(mbk_ctx *ctx, const char *addr, uint16_t port, uint64_t max_tiles, uint32_t senders,
                   mbk_worker_report *report)
{
    if (!ctx || !addr) return fail(ctx, MBK_ERR_INVALID, "NULL argument");
    for (Slot &sl : ctx->s)
        if (sl.busy) return fail(ctx, MBK_ERR_INVALID, "a slot still has a tile in flight: call mbk_wait first");
    MBK_HIP(ctx, hipSetDevice(ctx->device));
    mbk_feeder_ops ops;
    ops.user = ctx;
    ops.submit = ctx_submit;
    ops.wait = ctx_wait;
    ops.alloc = ctx_alloc;
    ops.release = ctx_release;
    ops.on_tile = nullptr;
    std::string err;
    const int rc = mbkf::run(&ops, addr, port, max_tiles, senders, report, &err, (uint32_t)MBK_WORKER_DEPTH);
    if (rc != MBK_OK) {
        // a backend failure left its own message on the ctx; keep it, prefixed
        const std::string inner = ctx->err;
        return fail(ctx, rc, inner.empty() || rc == MBK_ERR_NET ? err : err + ": " + inner);
    }
    return MBK_OK;
}

}  // extern "C"

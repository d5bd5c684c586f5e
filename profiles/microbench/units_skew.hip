// units_skew.hip -- diagnostic (round 4): can a launch be shortened by dealing the XCDs UNEQUAL shares of the heavy units?
//
// units_trace.hip showed the XCDs of one chip finishing equal static shares 5-10 % apart (profiles/NOTES.md 2b): the hardware
// deals workgroup ids to the XCDs in turn (id mod 8), every XCD works through its own ids, the launch lasts as long as its
// slowest XCD.  Cross-XCD atomics are too slow to rebalance at run time (units_pool_ab.txt), and moving only light units did
// not pay (units_weighted_deal_ab.txt).  Here the H units themselves are dealt by weight: XCD x (= id mod 8) takes h_x of the
// H list and S - h_x of the light list (S = ids per XCD), both through a closed-form rank (no atomics):
//     index(x, j) = sum_y min(j, c_y) + #{y < x : c_y > j}         (= 8 j + x while j < min c)
// The signal for the weights is what a product launch could afford: the START stamp of the last few workgroups of every
// XCD (the moment its dispatcher ran dry), 64 plain stores per launch.  Even / weighted launches alternate on the same box;
// the controller takes a speed estimate (share / time) from every launch.
// Same device code as the product (classify_units_kernel, block_pixel, escape_light_row from csrc/).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o build/units_skew profiles/microbench/units_skew.hip
//   build/units_skew cfg2 40 [gain 0.5] [rotation 0] [cycle test 0|1] [signal 0 = dry stamps | 1 = end of the XCD's last wave]
// (signal 1 is a diagnostic -- every workgroup stores its end time, the host takes the maximum per XCD -- for the question the
// product left open: with the cycle test a launch ends with the drain of its boundary blocks, which the dry stamps do not see.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../distributedmandelbrot_amd/csrc/mbk_units.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr uint32_t kTail = 8;   // stamps per XCD

struct Plan {
    uint32_t rot;       // diagnostic: XCD x takes the list positions of XCD (x + rot) mod 8 in the evenly dealt part
    uint32_t h[8];      // H units of XCD x
    uint32_t l[8];      // light units (M, then V) of XCD x
    uint32_t hmin, lmin, n_h, n_m, n_v, slots;   // slots: workgroup ids per XCD
};

__device__ __forceinline__ uint32_t rank8(const uint32_t *c, uint32_t j, uint32_t x)
{
    uint32_t r = 0;
#pragma unroll
    for (uint32_t y = 0; y < 8u; ++y) {
        r += c[y] < j ? c[y] : j;
        r += (y < x && c[y] > j) ? 1u : 0u;
    }
    return r;
}

template <bool kCycle>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void skew_units_kernel(mbk::TileArgs args, uint32_t qtab, Plan pl,
                                                                                                 unsigned long long *stamps, uint32_t *xcc_of,
                                                                                                 unsigned long long *ends)
{
    const unsigned long long t0 = wall_clock64();
    mbk::TileArgs p = args;
    p.smooth = nullptr; p.stats = nullptr; p.quant_wide = 0u; p.re.step_is_zero = p.im.step_is_zero = 0u; p.bytes = nullptr;
    const uint32_t lane = threadIdx.x, lx = lane & 7u, ly = lane >> 3;
    const uint32_t n = p.ngrid;
    const uint32_t b = blockIdx.x, x = b & 7u, j = b >> 3;
    if (lane == 0 && j + kTail >= pl.slots) stamps[8u + x * kTail + (j + kTail - pl.slots)] = t0;
    if (b < 8u && lane == 0) { stamps[b] = t0; xcc_of[b] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15u; }
    const uint32_t hx = pl.h[x];
    if (j < hx) {
        const uint32_t i = j < pl.hmin ? 8u * j + ((x + pl.rot) & 7u) : rank8(pl.h, j, x);
        const uint32_t e = mbk::uniform_u32(p.order[i]);
        const uint32_t by = e >> 16, bx = e & 0xffffu;
        mbk::block_pixel<double, true, 16, kCycle>(p, bx * 8u, by * 8u, lx, ly, true, bx < p.fast_bx_end && by < p.fast_by_end);
        if (lane == 0) ends[b] = wall_clock64();
        return;
    }
    const uint32_t k = j - hx;
    if (k >= pl.l[x]) { if (lane == 0) ends[b] = 0ull; return; }
    const uint32_t i = k < pl.lmin ? 8u * k + ((x + pl.rot) & 7u) : rank8(pl.l, k, x);
    if (i < pl.n_m) {
        const uint32_t e = mbk::uniform_u32(p.order[n + 3u + i]);
        const uint32_t by = e >> 16, bx = e & 0xffffu;
        mbk::block_pixel<double, true, 16, kCycle>(p, bx * 8u, by * 8u, lx, ly, false, bx < p.fast_bx_end && by < p.fast_by_end);
    } else {
        const uint32_t v = mbk::uniform_u32(p.order[n - 1u - (i - pl.n_m)]);
        const uint32_t by = v >> 16, bx0 = ((v >> 8) & 0xffu) << 3, mask = v & 0xffu;
        const double ci = (double)(p.row0 + by * 8u + ly) * p.im.step + p.im.start, b0 = ci * ci;
        const size_t elem0 = (size_t)(by * 8u + p.out_row0) * p.out_pitch + bx0 * 8u + p.out_col0;
        int32_t *cb = reinterpret_cast<int32_t *>(mbk::uniform_u64(reinterpret_cast<unsigned long long>(p.counts + elem0)));
        const uint32_t col = p.col0 + bx0 * 8u + lx, off = (ly * p.out_pitch + lx) * 4u;
        uint32_t kk = 0;
        int32_t cnt;
        while (mbk::escape_light_row<true, false>(ci, b0, col, p.re.step, p.re.start, cnt, cb, nullptr, off, 32u, qtab, mask, kk) != 0u) {
            mbk::block_pixel<double, true, 16, kCycle>(p, (bx0 + kk) * 8u, by * 8u, lx, ly, false, true);
            if (++kk >= 8u) break;
        }
    }
    if (lane == 0) ends[b] = wall_clock64();
}

// the shares of a launch from the fractions f (sum 1): h by cumulative rounding, light = what is left of S ids per XCD
static Plan make_plan(const double *f, uint32_t n_h, uint32_t n_m, uint32_t n_v)
{
    Plan pl; memset(&pl, 0, sizeof(pl));
    pl.n_h = n_h; pl.n_m = n_m; pl.n_v = n_v;
    const uint32_t total = n_h + n_m + n_v, n_l = n_m + n_v;
    double cum = 0; uint32_t prev = 0;
    for (int x = 0; x < 8; ++x) {
        cum += f[x];
        const uint32_t c = x == 7 ? n_h : (uint32_t)std::llround(cum * n_h);
        pl.h[x] = c - prev; prev = c;
    }
    uint32_t S = (total + 7) / 8;
    for (int x = 0; x < 8; ++x) S = std::max(S, pl.h[x]);
    // light shares: fill every XCD up to S ids; the (< 8) ids too many come off the last XCDs
    long long extra = 0;
    for (int x = 0; x < 8; ++x) { pl.l[x] = S - pl.h[x]; extra += pl.l[x]; }
    extra -= n_l;
    if (extra < 0) {   // the H shares are so uneven that S ids per XCD do not hold the light list: more ids
        const uint32_t add = (uint32_t)((-extra + 7) / 8);
        S += add;
        for (int x = 0; x < 8; ++x) pl.l[x] += add;
        extra += 8ll * add;
    }
    for (int x = 7; extra > 0; x = (x + 7) % 8) { if (pl.l[x]) { --pl.l[x]; --extra; } }
    pl.slots = S;
    pl.hmin = *std::min_element(pl.h, pl.h + 8);
    pl.lmin = *std::min_element(pl.l, pl.l + 8);
    return pl;
}

int main(int argc, char **argv)
{
    const std::string wl = argc > 1 ? argv[1] : "cfg2";
    const int reps = argc > 2 ? atoi(argv[2]) : 40;
    const double gain = argc > 3 ? atof(argv[3]) : 0.5;
    const uint32_t rot = argc > 4 ? (uint32_t)atoi(argv[4]) : 0u;
    const bool cyc = argc > 5 && atoi(argv[5]) != 0;
    const int signal = argc > 6 ? atoi(argv[6]) : 0;
    const uint32_t W = 4096, H = 4096, mrd = 1000;
    mbk::TileArgs a; memset(&a, 0, sizeof(a));
    auto mk = [](double start, double range, uint32_t n) { mbk::Axis x; memset(&x, 0, sizeof(x)); x.start = start; x.n = n;
        volatile double stop = start + range, delta = stop - start, div = n - 1, step = delta / div;
        x.last = stop; x.delta = delta; x.div = div; x.step = step; x.step_is_zero = 0; return x; };
    if (wl == "chunk_l1") { a.re = mk(-2.0, 4.0, W); a.im = mk(-2.0, 4.0, H); }
    else { a.re = mk(-2.0, 3.0, W); a.im = mk(-1.5, 3.0, H); }
    a.ncols = W; a.nrows = H; a.out_pitch = W; a.mrd = mrd; a.quant_rcp = 1.0 / mrd;
    a.exact_steps = 8; a.exact_steps_long = 0; a.ring_possible = 1;
    a.blocks_x = W / 8;
    a.fast_bx_end = W / 8 - 1; a.fast_by_end = H / 8 - 1;
    a.perm_mul = 1;
    const uint32_t nblocks = a.blocks_x * (H / 8);
    CHECK(hipMalloc(&a.counts, (size_t)W * H * 4));
    int32_t *ref_counts; CHECK(hipMalloc(&ref_counts, (size_t)W * H * 4));
    uint32_t *ord; CHECK(hipMalloc(&ord, mbk::units_list_words(nblocks) * 4));
    a.order = ord; a.ngrid = nblocks; a.unit_stride = nblocks;
    unsigned long long *stamps; CHECK(hipHostMalloc(&stamps, (8 + 8 * kTail) * 8, hipHostMallocDefault));
    unsigned long long *ends; CHECK(hipMalloc(&ends, (size_t)nblocks * 8));
    std::vector<unsigned long long> hends(nblocks);
    uint32_t *xcc_of; CHECK(hipHostMalloc(&xcc_of, 8 * 4, hipHostMallocDefault));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    uint32_t cnt[3] = {0, 0, 0};
    CHECK(hipMemset(ord + nblocks, 0, 12));
    mbk::classify_units_kernel<<<(nblocks + 1023) / 1024, 1024>>>(a, nblocks, 32, ord, ord + nblocks, 0, 0.0, ord + 2 * (size_t)nblocks + 3, nullptr, mbk::XcdShares(), 0u);
    CHECK(hipMemcpy(cnt, ord + nblocks, 12, hipMemcpyDeviceToHost));
    const uint32_t n_h = cnt[0], n_v = cnt[1], n_m = cnt[2];
    printf("%s: H %u, V units %u, M %u; controller gain %.2f, rotation %u, cycle test %d, signal %s\n", wl.c_str(), n_h, n_v, n_m, gain, rot, (int)cyc,
           signal ? "end of the last wave per XCD" : "dry stamps");
    {   // the clock ramps for the first ~100 launches after an idle period: get that out of the way
        const double ev[8] = {0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125, 0.125};
        const Plan pw = make_plan(ev, n_h, n_m, n_v);
        for (int w = 0; w < 150; ++w) {
            if (cyc) skew_units_kernel<true><<<pw.slots * 8, 64>>>(a, 0u, pw, stamps, xcc_of, ends);
            else skew_units_kernel<false><<<pw.slots * 8, 64>>>(a, 0u, pw, stamps, xcc_of, ends);
        }
        CHECK(hipDeviceSynchronize());
    }

    double f[8], even[8];
    for (int x = 0; x < 8; ++x) f[x] = even[x] = 0.125;
    std::vector<double> ms_even, ms_w;
    std::vector<int32_t> href((size_t)W * H), hnow((size_t)W * H);
    for (int rep = 0; rep < reps + 4; ++rep) {
        const bool weighted = rep >= 4 && (rep & 1);      // four even launches to warm up, then alternate
        Plan pl = make_plan(weighted ? f : even, n_h, n_m, n_v);
        pl.rot = rot;
        memset(stamps, 0, (8 + 8 * kTail) * 8);
        if (rep == reps + 3 || rep == reps + 2) CHECK(hipMemset(a.counts, 0xff, (size_t)W * H * 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        if (cyc) skew_units_kernel<true><<<pl.slots * 8, 64>>>(a, 0u, pl, stamps, xcc_of, ends);
        else skew_units_kernel<false><<<pl.slots * 8, 64>>>(a, 0u, pl, stamps, xcc_of, ends);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long start = ~0ull;
        for (int x = 0; x < 8; ++x) if (stamps[x] && stamps[x] < start) start = stamps[x];
        double T[8];
        for (int x = 0; x < 8; ++x) {
            unsigned long long last = 0;
            for (uint32_t t = 0; t < kTail; ++t) last = std::max(last, stamps[8 + x * kTail + t]);
            T[x] = (double)(last - start) / 100.0;   // us
        }
        // the other signal: when the last wave of every XCD ended
        double E[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        CHECK(hipMemcpy(hends.data(), ends, (size_t)pl.slots * 8 * 8, hipMemcpyDeviceToHost));
        for (uint32_t b = 0; b < pl.slots * 8; ++b)
            if (hends[b] > start) E[b & 7] = std::max(E[b & 7], (double)(hends[b] - start) / 100.0);
        printf("%s rep %2d %s %.4f ms  ids/XCD %u  dry at us:", wl.c_str(), rep, weighted ? "W" : "E", ms, pl.slots);
        for (int x = 0; x < 8; ++x) printf(" %6.1f", T[x]);
        printf("  ended at:");
        for (int x = 0; x < 8; ++x) printf(" %6.1f", E[x]);
        if (signal) for (int x = 0; x < 8; ++x) T[x] = E[x];
        printf("  h:");
        for (int x = 0; x < 8; ++x) printf(" %u", pl.h[x]);
        printf("\n");
        if (rep >= 4) (weighted ? ms_w : ms_even).push_back(ms);
        // speed estimate of this launch: share / time; new fractions proportional to it, damped
        double s[8], sum = 0;
        for (int x = 0; x < 8; ++x) { s[x] = (double)pl.h[x] / std::max(T[x], 1.0); sum += s[x]; }
        if (rep >= 2) for (int x = 0; x < 8; ++x) f[x] = (1.0 - gain) * f[x] + gain * s[x] / sum;
        double fs = 0; for (int x = 0; x < 8; ++x) fs += f[x];
        for (int x = 0; x < 8; ++x) f[x] /= fs;
        if (rep == reps + 2) CHECK(hipMemcpy(href.data(), a.counts, (size_t)W * H * 4, hipMemcpyDeviceToHost));   // even
        if (rep == reps + 3) {                                                                                      // weighted
            CHECK(hipMemcpy(hnow.data(), a.counts, (size_t)W * H * 4, hipMemcpyDeviceToHost));
            size_t bad = 0, unset = 0;
            for (size_t i = 0; i < href.size(); ++i) { bad += href[i] != hnow[i]; unset += hnow[i] == -1; }
            printf("%s: weighted launch against even launch: %zu differing counts, %zu pixels never written\n", wl.c_str(), bad, unset);
        }
    }
    printf("%s: id mod 8 -> XCC_ID:", wl.c_str());
    for (int x = 0; x < 8; ++x) printf(" %u", xcc_of[x]);
    auto stat = [](std::vector<double> v, size_t from) { std::vector<double> w(v.begin() + std::min(from, v.size()), v.end()); std::sort(w.begin(), w.end());
        double m = 0; for (double q : w) m += q; return std::pair<double, double>(w.empty() ? 0 : m / w.size(), w.empty() ? 0 : w[w.size() / 2]); };
    const auto se = stat(ms_even, ms_even.size() / 2), sw = stat(ms_w, ms_w.size() / 2);
    printf("\n%s: second half of the run: even mean %.4f median %.4f ms; weighted mean %.4f median %.4f ms -> weighted/even %.4f (median)\n",
           wl.c_str(), se.first, se.second, sw.first, sw.second, sw.second / se.second);
    printf("%s: final fractions:", wl.c_str());
    for (int x = 0; x < 8; ++x) printf(" %.4f", f[x]);
    printf("\n");
    return 0;
}

// units_classes.hip -- diagnostic (round 6): where do the instructions of a cfg2 "units" launch go, class by class?
// The product's dispatch list (classify_units_kernel) and the product's routines (block_pixel, escape_light_row from csrc/),
// but every class of the list -- late M, H, settled H, M, V units -- is launched as a kernel OF ITS OWN (the class is a template
// argument, so the kernel names differ), next to the launch of the whole list: `rocprofv3 --pmc` of this binary then gives VALU
// instructions, busy cycles and lane activity per class.  The host adds the exact work of every class from the counts the
// launch wrote (pixel-iterations; the ideal instruction count is 6.125 / 64 of that for 16-step groups, 6.25 / 64 for 8-step).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o /tmp/units_classes profiles/microbench/units_classes.hip
//   /tmp/units_classes cfg2|chunk_l1 [cycle test 0|1] [m_late 8] [h_settled 6]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../distributedmandelbrot_amd/csrc/mbk_units.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// kClass: 0 the whole list in the product's order (late M, H, settled H, M, V units); 1 late M; 2 H; 3 settled H; 4 M; 5 V units.
// `u0`: where the class begins in that order (the grid is the class's size).
template <bool kCycle, int kClass>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void class_units_kernel(mbk::TileArgs args, uint32_t qtab, uint32_t u0)
{
    mbk::TileArgs p = args;
    p.smooth = nullptr; p.stats = nullptr; p.quant_wide = 0u; p.re.step_is_zero = p.im.step_is_zero = 0u; p.bytes = nullptr;
    const uint32_t lane = threadIdx.x, lx = lane & 7u, ly = lane >> 3;
    const uint32_t n = p.ngrid;
    const uint32_t n_h = mbk::uniform_u32(p.order[n]), n_v = mbk::uniform_u32(p.order[n + 1u]), n_m = mbk::uniform_u32(p.order[n + 2u]);
    const uint32_t n_ml = mbk::uniform_u32(p.order[2u * n + 3u]), n_hs = mbk::uniform_u32(p.order[2u * n + 4u]);
    const uint32_t u = blockIdx.x + u0;
    if (u < n_ml + n_h + n_hs + n_m) {
        const bool late = u < n_ml, is_hu = !late && u < n_ml + n_h, is_hs = !late && !is_hu && u < n_ml + n_h + n_hs;
        const uint32_t e = mbk::uniform_u32(late ? p.order[2u * n + 2u - u] : is_hu ? p.order[u - n_ml]
                                            : is_hs ? p.order[mbk::units_settled_base(n) + (u - n_ml - n_h)] : p.order[n + 3u + (u - n_ml - n_h - n_hs)]);
        const uint32_t by = e >> 16, bx = e & 0xffffu;
        mbk::block_pixel<double, true, 16, kCycle>(p, bx * 8u, by * 8u, lx, ly, is_hu || is_hs, bx < p.fast_bx_end && by < p.fast_by_end);
    } else if (u < n_ml + n_h + n_hs + n_m + n_v) {
        const uint32_t v = mbk::uniform_u32(p.order[n - 1u - (u - n_ml - n_h - n_hs - n_m)]);
        const uint32_t by = v >> 16, bx0 = ((v >> 8) & 0xffu) << 3, mask = v & 0xffu;
        const double ci = (double)(p.row0 + by * 8u + ly) * p.im.step + p.im.start, b0 = ci * ci;
        const size_t elem0 = (size_t)(by * 8u + p.out_row0) * p.out_pitch + bx0 * 8u + p.out_col0;
        int32_t *cb = reinterpret_cast<int32_t *>(mbk::uniform_u64(reinterpret_cast<unsigned long long>(p.counts + elem0)));
        const uint32_t col = p.col0 + bx0 * 8u + lx, off = (ly * p.out_pitch + lx) * 4u;
        uint32_t k = 0;
        int32_t cnt;
        while (mbk::escape_light_row<true, false>(ci, b0, col, p.re.step, p.re.start, cnt, cb, nullptr, off, 32u, qtab, mask, k) != 0u) {
            mbk::block_pixel<double, true, 16, kCycle>(p, (bx0 + k) * 8u, by * 8u, lx, ly, false, true);
            if (++k >= 8u) break;
        }
    }
}

template <bool kCycle>
static void launch_class(int cls, uint32_t grid, const mbk::TileArgs &a, uint32_t u0)
{
    if (grid == 0) return;
    switch (cls) {
        case 0: class_units_kernel<kCycle, 0><<<grid, 64>>>(a, 0u, u0); break;
        case 1: class_units_kernel<kCycle, 1><<<grid, 64>>>(a, 0u, u0); break;
        case 2: class_units_kernel<kCycle, 2><<<grid, 64>>>(a, 0u, u0); break;
        case 3: class_units_kernel<kCycle, 3><<<grid, 64>>>(a, 0u, u0); break;
        case 4: class_units_kernel<kCycle, 4><<<grid, 64>>>(a, 0u, u0); break;
        default: class_units_kernel<kCycle, 5><<<grid, 64>>>(a, 0u, u0); break;
    }
}

int main(int argc, char **argv)
{
    const std::string wl = argc > 1 ? argv[1] : "cfg2";
    const bool cyc = argc > 2 && atoi(argv[2]) != 0;
    const int m_late = argc > 3 ? atoi(argv[3]) : 8;
    const int h_settled = cyc ? (argc > 4 ? atoi(argv[4]) : 6) : 0;
    const double settle_thr = h_settled ? pow(10.0, -(double)h_settled) : 0.0;
    const uint32_t W = 4096, H = 4096, mrd = 1000;
    mbk::TileArgs a; memset(&a, 0, sizeof(a));
    auto mk = [](double start, double range, uint32_t n) { mbk::Axis x; memset(&x, 0, sizeof(x)); x.start = start; x.n = n;
        volatile double stop = start + range, delta = stop - start, div = n - 1, step = delta / div;
        x.last = stop; x.delta = delta; x.div = div; x.step = step; x.step_is_zero = 0; return x; };
    if (wl == "chunk_l1") { a.re = mk(-2.0, 4.0, W); a.im = mk(-2.0, 4.0, H); }
    else { a.re = mk(-2.0, 3.0, W); a.im = mk(-1.5, 3.0, H); }
    a.ncols = W; a.nrows = H; a.out_pitch = W; a.mrd = mrd; a.quant_rcp = 1.0 / mrd;
    a.exact_steps = 8; a.exact_steps_long = 0; a.ring_possible = 1; a.cyc_window = 32;
    a.blocks_x = W / 8;
    a.fast_bx_end = W / 8 - 1; a.fast_by_end = H / 8 - 1;
    a.perm_mul = 1;
    const uint32_t nblocks = a.blocks_x * (H / 8);
    CHECK(hipMalloc(&a.counts, (size_t)W * H * 4));
    uint32_t *ord; CHECK(hipMalloc(&ord, mbk::units_list_words(nblocks) * 4));
    a.order = ord; a.ngrid = nblocks; a.unit_stride = nblocks;
    CHECK(hipMemset(ord + nblocks, 0, 12));
    CHECK(hipMemset(ord + 2 * (size_t)nblocks + 3, 0, 8));
    mbk::classify_units_kernel<<<(nblocks + 1023) / 1024, 1024>>>(a, nblocks, 32, ord, ord + nblocks, m_late, settle_thr, ord + 2 * (size_t)nblocks + 3, nullptr, mbk::XcdShares(), 0u);
    std::vector<uint32_t> lst(mbk::units_list_words(nblocks));
    CHECK(hipMemcpy(lst.data(), ord, lst.size() * 4, hipMemcpyDeviceToHost));
    const uint32_t n = nblocks, n_h = lst[n], n_v = lst[n + 1], n_m = lst[n + 2], n_ml = lst[2 * (size_t)n + 3], n_hs = lst[2 * (size_t)n + 4];
    const uint32_t size[6] = {n_ml + n_h + n_hs + n_m + n_v, n_ml, n_h, n_hs, n_m, n_v};
    const uint32_t first[6] = {0, 0, n_ml, n_ml + n_h, n_ml + n_h + n_hs, n_ml + n_h + n_hs + n_m};
    const char *names[6] = {"all", "late M", "H", "settled H", "M", "V units"};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 150; ++rep) {                       // the clock
        if (cyc) launch_class<true>(0, size[0], a, 0); else launch_class<false>(0, size[0], a, 0);
    }
    CHECK(hipDeviceSynchronize());
    double ms[6] = {0, 0, 0, 0, 0, 0};
    const int reps = 5;
    for (int rep = 0; rep < reps; ++rep)
        for (int c = 0; c < 6; ++c) {
            // (a class alone: the same blocks through the same routines; its duration alone is NOT its share of the whole launch --
            // the V units alone are bound by the dispatcher, a class of long blocks by its own drain)
            for (int w = 0; w < 2; ++w) { if (cyc) launch_class<true>(0, size[0], a, 0); else launch_class<false>(0, size[0], a, 0); }
            CHECK(hipEventRecord(e0));
            if (cyc) launch_class<true>(c, size[c], a, first[c]); else launch_class<false>(c, size[c], a, first[c]);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float t = 0; if (size[c]) CHECK(hipEventElapsedTime(&t, e0, e1));
            ms[c] += t / reps;
        }
    // exact work per class from the counts the launches wrote
    std::vector<int32_t> cnt((size_t)W * H);
    CHECK(hipMemcpy(cnt.data(), a.counts, cnt.size() * 4, hipMemcpyDeviceToHost));
    auto block_work = [&](uint32_t bx, uint32_t by, unsigned long long &iters, unsigned long long &wave_steps) {
        uint32_t mx = 0;
        for (uint32_t y = 0; y < 8; ++y) for (uint32_t x = 0; x < 8; ++x) {
            const int32_t c = cnt[(size_t)(by * 8 + y) * W + bx * 8 + x];
            const uint32_t it = c > 0 ? (uint32_t)c : mrd - 1;
            iters += it; mx = it > mx ? it : mx;
        }
        wave_steps += mx;
    };
    unsigned long long iters[6] = {0, 0, 0, 0, 0, 0}, wsteps[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t u = 0; u < size[0]; ++u) {
        int c;
        if (u < first[2]) c = 1; else if (u < first[3]) c = 2; else if (u < first[4]) c = 3; else if (u < first[5]) c = 4; else c = 5;
        if (c < 5) {
            const uint32_t e = c == 1 ? lst[2 * (size_t)n + 2 - u] : c == 2 ? lst[u - n_ml] : c == 3 ? lst[mbk::units_settled_base(n) + (u - n_ml - n_h)]
                                                                                              : lst[n + 3 + (u - n_ml - n_h - n_hs)];
            block_work(e & 0xffffu, e >> 16, iters[c], wsteps[c]);
        } else {
            const uint32_t v = lst[n - 1 - (u - first[5])];
            const uint32_t by = v >> 16, bx0 = ((v >> 8) & 0xffu) << 3, mask = v & 0xffu;
            for (uint32_t k = 0; k < 8; ++k) if (mask >> k & 1u) block_work(bx0 + k, by, iters[c], wsteps[c]);
        }
    }
    for (int c = 1; c < 6; ++c) { iters[0] += iters[c]; wsteps[0] += wsteps[c]; }
    printf("%s, cycle test %d, m_late %d, h_settled %d: strict work per class from the counts (a pixel's iterations = count, or mrd - 1 = %u for a never-escaping one; "
           "wave-steps = the block's longest pixel, what a lock-step wave runs WITHOUT the cycle test)\n", wl.c_str(), (int)cyc, m_late, h_settled, mrd - 1);
    printf("%-10s %9s %16s %14s %10s %12s\n", "class", "entries", "pixel-iterations", "wave-steps", "activity", "ms alone");
    for (int c = 0; c < 6; ++c)
        printf("%-10s %9u %16llu %14llu %10.4f %12.4f\n", names[c], size[c], iters[c], wsteps[c], wsteps[c] ? (double)iters[c] / 64.0 / (double)wsteps[c] : 0.0, ms[c]);
    return 0;
}

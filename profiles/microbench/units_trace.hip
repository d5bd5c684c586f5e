// units_trace.hip -- diagnostic (round 4): the product's "units" launch of BASELINE cfg2 (or DataChunk (1,0,0)) with a time
// stamp pair and the hardware placement of every workgroup, to SEE where the idle issue slots of a launch are: the ramp,
// the drain of the last waves, imbalance between XCDs / SIMDs.  Same device code as the product (classify_units_kernel,
// block_pixel, escape_light_row from csrc/); the traced kernel is tile_units_kernel's body with the grid sized exactly (the
// host reads the counters) plus the record.  One record per workgroup:
//   u64 t0, t1 (s_memrealtime, 100 MHz), u32 hw_id, u32 xcc_id, u32 unit index, u32 class (0 H, 1 M, 2 V)
// Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o /tmp/units_trace profiles/microbench/units_trace.hip
//   /tmp/units_trace cfg2 gpurun_out/units_trace_cfg2.bin && python scripts/analyze_units_trace.py gpurun_out/units_trace_cfg2.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../distributedmandelbrot_amd/csrc/mbk_units.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Rec { unsigned long long t0, t1; unsigned hw, xcc, unit, cls; };

// Dispatch order: [late M entries (round 5, MBK_OPT_M_LATE: from the back of the M region)] [H] [settled H (MBK_OPT_H_SETTLED: a
// region of their own)] [M] [V units]; class codes in the record: 0 H, 1 M, 2 V, 4 late M, 5 settled H.
template <bool kCycle>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void traced_units_kernel(mbk::TileArgs args, uint32_t qtab, Rec *rec)
{
    const unsigned long long t0 = wall_clock64();
    mbk::TileArgs p = args;
    p.smooth = nullptr; p.stats = nullptr; p.quant_wide = 0u; p.re.step_is_zero = p.im.step_is_zero = 0u; p.bytes = nullptr;
    const uint32_t lane = threadIdx.x, lx = lane & 7u, ly = lane >> 3;
    const uint32_t n = p.ngrid;
    const uint32_t n_h = mbk::uniform_u32(p.order[n]), n_v = mbk::uniform_u32(p.order[n + 1u]), n_m = mbk::uniform_u32(p.order[n + 2u]);
    const uint32_t n_ml = mbk::uniform_u32(p.order[2u * n + 3u]), n_hs = mbk::uniform_u32(p.order[2u * n + 4u]);
    const uint32_t u = blockIdx.x;
    uint32_t cls = 3u;
    if (u < n_ml + n_h + n_hs + n_m) {
        const bool late = u < n_ml, is_hu = !late && u < n_ml + n_h, is_hs = !late && !is_hu && u < n_ml + n_h + n_hs;
        cls = late ? 4u : is_hu ? 0u : is_hs ? 5u : 1u;
        const uint32_t e = mbk::uniform_u32(late ? p.order[2u * n + 2u - u] : is_hu ? p.order[u - n_ml]
                                            : is_hs ? p.order[mbk::units_settled_base(n) + (u - n_ml - n_h)] : p.order[n + 3u + (u - n_ml - n_h - n_hs)]);
        const uint32_t by = e >> 16, bx = e & 0xffffu;
        mbk::block_pixel<double, true, 16, kCycle>(p, bx * 8u, by * 8u, lx, ly, is_hu || is_hs, bx < p.fast_bx_end && by < p.fast_by_end);
    } else if (u < n_ml + n_h + n_hs + n_m + n_v) {
        cls = 2u;
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
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) {
        Rec r; r.t0 = t0; r.t1 = t1;
        r.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        r.unit = u; r.cls = cls;
        rec[u] = r;
    }
}

int main(int argc, char **argv)
{
    const std::string wl = argc > 1 ? argv[1] : "cfg2";
    const char *out = argc > 2 ? argv[2] : "gpurun_out/units_trace.bin";
    const bool cyc = argc > 3 && atoi(argv[3]) != 0;          // the library's default: the cycle test
    const int m_late = argc > 4 ? atoi(argv[4]) : 0;          // MBK_OPT_M_LATE
    const int h_settled = argc > 5 ? atoi(argv[5]) : 0;       // MBK_OPT_H_SETTLED (k: threshold 10^-k)
    const uint32_t cyc_window = argc > 6 ? (uint32_t)atoi(argv[6]) : 32u;   // MBK_OPT_CYCLE_WINDOW (the library's default)
    const double settle_thr = h_settled ? pow(10.0, -(double)h_settled) : 0.0;
    const uint32_t W = 4096, H = 4096, mrd = 1000;
    mbk::TileArgs a; memset(&a, 0, sizeof(a));
    auto mk = [](double start, double range, uint32_t n) { mbk::Axis x; memset(&x, 0, sizeof(x)); x.start = start; x.n = n;
        volatile double stop = start + range, delta = stop - start, div = n - 1, step = delta / div;
        x.last = stop; x.delta = delta; x.div = div; x.step = step; x.step_is_zero = 0; return x; };
    if (wl == "chunk_l1") { a.re = mk(-2.0, 4.0, W); a.im = mk(-2.0, 4.0, H); }
    else { a.re = mk(-2.0, 3.0, W); a.im = mk(-1.5, 3.0, H); }
    a.ncols = W; a.nrows = H; a.out_pitch = W; a.mrd = mrd; a.quant_rcp = 1.0 / mrd;
    a.exact_steps = 8; a.exact_steps_long = 0; a.ring_possible = 1; a.cyc_window = cyc_window;
    a.blocks_x = W / 8;
    a.fast_bx_end = W / 8 - 1; a.fast_by_end = H / 8 - 1;   // (conservative: the blocks holding an axis' last sample take the general path)
    a.perm_mul = 1;
    const uint32_t nblocks = a.blocks_x * (H / 8);
    CHECK(hipMalloc(&a.counts, (size_t)W * H * 4));
    uint32_t *ord; CHECK(hipMalloc(&ord, mbk::units_list_words(nblocks) * 4));
    a.order = ord; a.ngrid = nblocks; a.unit_stride = nblocks;
    Rec *d; CHECK(hipMalloc(&d, (size_t)nblocks * sizeof(Rec)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    uint32_t cnt[3] = {0, 0, 0}, n_ml = 0, n_hs = 0;
    for (int rep = 0; rep < 160; ++rep) {      // (the clock needs ~100 launches to settle; the last launch is the one recorded)
        CHECK(hipMemset(ord + nblocks, 0, 12));
        CHECK(hipMemset(ord + 2 * (size_t)nblocks + 3, 0, 8));
        mbk::classify_units_kernel<<<(nblocks + 1023) / 1024, 1024>>>(a, nblocks, 32, ord, ord + nblocks, m_late, settle_thr, ord + 2 * (size_t)nblocks + 3, nullptr, mbk::XcdShares(), 0u);
        CHECK(hipMemcpy(cnt, ord + nblocks, 12, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&n_ml, ord + 2 * (size_t)nblocks + 3, 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&n_hs, ord + 2 * (size_t)nblocks + 4, 4, hipMemcpyDeviceToHost));
        const uint32_t total = cnt[0] + cnt[1] + cnt[2] + n_ml + n_hs;
        CHECK(hipMemset(d, 0, (size_t)nblocks * sizeof(Rec)));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        if (cyc) traced_units_kernel<true><<<total, 64>>>(a, 0u, d);
        else traced_units_kernel<false><<<total, 64>>>(a, 0u, d);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 156) printf("%s rep %d (cycle test %d, m_late %d, h_settled %d, cycle window %u): late M %u, H %u, settled H %u, V units %u, M %u -> %u workgroups, %.3f ms\n", wl.c_str(), rep, (int)cyc, m_late, h_settled, cyc_window, n_ml, cnt[0], n_hs, cnt[1], cnt[2], total, ms);
    }
    const uint32_t total = cnt[0] + cnt[1] + cnt[2] + n_ml + n_hs;
    std::vector<Rec> h(total);
    CHECK(hipMemcpy(h.data(), d, (size_t)total * sizeof(Rec), hipMemcpyDeviceToHost));
    FILE *f = fopen(out, "wb"); fwrite(h.data(), sizeof(Rec), total, f); fclose(f);
    printf("wrote %u records to %s\n", total, out);
    return 0;
}

// occupancy_trace.hip -- diagnostic: runs the production escape loop (escape_count_asm from
// mbk_kernels.h) on BASELINE cfg2 with per-wave timestamps and hardware placement, and writes
// gpurun_out/trace_<tag>.bin for scripts/analyze_trace.py.  One record per wave:
//   u64 t_start, u64 t_end (wall_clock64, 100 MHz), u32 hw_id, u32 xcc_id, u32 block, u32 count_sum
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o /tmp/occ profiles/microbench/occupancy_trace.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <utility>
#include "../../distributedmandelbrot_amd/csrc/mbk_kernels.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Rec { unsigned long long t0, t1; unsigned hw, xcc, blk, sum; };

template <int MODE>
__global__ __launch_bounds__(256) void traced_kernel(mbk::TileArgs p, Rec *rec, const uint32_t *order)
{
    const unsigned long long t0 = wall_clock64();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t blk = order ? order[blockIdx.x] : (uint32_t)(((uint64_t)blockIdx.x * p.perm_mul) % gridDim.x);
    const uint32_t by = blk / p.blocks_x, bx = blk - by * p.blocks_x;
    const uint32_t lc = (bx * (blockDim.x >> 6) + wave) * 8u + (lane & 7u), lr = by * 8u + (lane >> 3);
    int32_t count = 0;
    if (lc < p.ncols && lr < p.nrows) {
        const double cr = mbk::axis_value(p.re, p.col0 + lc), ci = mbk::axis_value(p.im, p.row0 + lr);
        count = MODE == 1 ? mbk::escape_count_group<8>(cr, ci, p.mrd) : mbk::escape_count_asm<true>(cr, ci, p.mrd);
        p.counts[(size_t)lr * p.ncols + lc] = count;
    }
    unsigned s = count > 0 ? (unsigned)count : (unsigned)(p.mrd - 1);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    const unsigned long long t1 = wall_clock64();
    if (lane == 0) {
        Rec r; r.t0 = t0; r.t1 = t1;
        r.hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        r.xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        r.blk = blk * (blockDim.x >> 6) + wave; r.sum = s;
        rec[blockIdx.x * (blockDim.x >> 6) + wave] = r;
    }
}

int main(int argc, char **argv)
{
    const int wpw = argc > 1 ? atoi(argv[1]) : 1;
    const int order = argc > 2 ? atoi(argv[2]) : 0;
    const char *out = argc > 3 ? argv[3] : "gpurun_out/trace.bin";
    const uint32_t W = 4096, H = 4096, mrd = 1000;
    mbk::TileArgs a; memset(&a, 0, sizeof(a));
    auto mk = [](double start, double range, uint32_t n) { mbk::Axis x; memset(&x, 0, sizeof(x)); x.start = start; x.n = n;
        volatile double stop = start + range, delta = stop - start, div = n - 1, step = delta / div;
        x.last = stop; x.delta = delta; x.div = div; x.step = step; x.step_is_zero = 0; return x; };
    a.re = mk(-2.0, 3.0, W); a.im = mk(-1.5, 3.0, H);
    a.ncols = W; a.nrows = H; a.mrd = mrd;
    a.blocks_x = (W + 8 * wpw - 1) / (8 * wpw);
    const uint32_t grid = a.blocks_x * ((H + 7) / 8);
    uint32_t mul = 1;
    if (order) { uint64_t m = (uint64_t)(grid * 0.6180339887498949); for (;; ++m) { uint64_t x = m, y = grid; while (y) { uint64_t t = x % y; x = y; y = t; } if (x == 1) break; } mul = (uint32_t)(m % grid); }
    a.perm_mul = mul;
    CHECK(hipMalloc(&a.counts, (size_t)W * H * 4));
    const size_t nw = (size_t)grid * wpw;
    Rec *d; CHECK(hipMalloc(&d, nw * sizeof(Rec)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        traced_kernel<0><<<grid, 64 * wpw>>>(a, d, nullptr);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("wpw %d order %d rep %d: %.3f ms\n", wpw, order, rep, ms);
    }
    std::vector<Rec> h(nw);
    CHECK(hipMemcpy(h.data(), d, nw * sizeof(Rec), hipMemcpyDeviceToHost));
    FILE *f = fopen(out, "wb"); fwrite(h.data(), sizeof(Rec), nw, f); fclose(f);
    printf("wrote %zu records to %s\n", nw, out);
    if (wpw == 1) {
        // production configuration: grouped loop + classify_blocks_kernel order
        uint32_t *dord2; CHECK(hipMalloc(&dord2, (nw + 2) * 4));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(dord2 + nw, 0, 8));
            CHECK(hipEventRecord(e0));
            mbk::classify_blocks_kernel<<<(grid + 1023) / 1024, 1024>>>(a, grid, 8, 32, dord2, dord2 + nw);
            traced_kernel<1><<<grid, 64>>>(a, d, dord2);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("group8 + classify order rep %d: %.3f ms\n", rep, ms);
        }
        CHECK(hipMemcpy(h.data(), d, nw * sizeof(Rec), hipMemcpyDeviceToHost));
        { FILE *g = fopen("gpurun_out/trace_w1_group.bin", "wb"); fwrite(h.data(), sizeof(Rec), nw, g); fclose(g); }
        CHECK(hipFree(dord2));
    }
    if (wpw == 1 && argc > 4) {
        // EXPERIMENT (oracle ordering from the previous run's own counts): heavy blocks first.
        std::vector<std::pair<unsigned, unsigned>> key(nw);
        for (size_t i = 0; i < nw; ++i) key[h[i].blk] = {h[i].sum, h[i].blk};
        std::vector<uint32_t> ord(nw);
        for (int mode = 0; mode < 3; ++mode) {
            std::vector<std::pair<unsigned, unsigned>> k2 = key;
            if (mode == 0) std::stable_sort(k2.begin(), k2.end(), [](auto &x, auto &y) { return x.first > y.first; });           // exact LPT
            if (mode == 1) std::stable_sort(k2.begin(), k2.end(), [](auto &x, auto &y) { return (x.first > 64 * 24) > (y.first > 64 * 24); });  // 2 classes
            if (mode == 2) std::stable_sort(k2.begin(), k2.end(), [](auto &x, auto &y) { return x.first < y.first; });           // worst case: trivial first
            for (size_t i = 0; i < nw; ++i) ord[i] = k2[i].second;
            uint32_t *dord; CHECK(hipMalloc(&dord, nw * 4)); CHECK(hipMemcpy(dord, ord.data(), nw * 4, hipMemcpyHostToDevice));
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                traced_kernel<0><<<grid, 64>>>(a, d, dord);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf("ordered mode %d (0 = LPT, 1 = heavy class first, 2 = trivial first) rep %d: %.3f ms\n", mode, rep, ms);
            }
            if (mode == 1) {
                CHECK(hipMemcpy(h.data(), d, nw * sizeof(Rec), hipMemcpyDeviceToHost));
                FILE *g = fopen("gpurun_out/trace_w1_heavyfirst.bin", "wb"); fwrite(h.data(), sizeof(Rec), nw, g); fclose(g);
            }
            CHECK(hipFree(dord));
        }
    }
    return 0;
}

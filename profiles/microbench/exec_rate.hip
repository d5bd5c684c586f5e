// exec_rate.hip -- round 6: does gfx950 skip the 16-lane passes of a wave64 fp64 instruction whose lanes are all masked off?
// A wave64 fp64 VALU instruction issues over 4 cycles (16 lanes per cycle).  If a pass whose 16 EXEC bits are zero were skipped, compacting a
// block's few live lanes into one quarter of the wave (a handful of ds_bpermute per compaction) would buy back most of what lock-step costs
// (lane activity 0.79-0.94 on the product's workloads) without leaving the wave.  Measured here: a chip full of waves (8 per SIMD) running a
// chain-free stream of v_fma_f64 under different EXEC masks; ns per wave-instruction per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_rate profiles/microbench/exec_rate.hip && /tmp/exec_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void fma_stream(double *sink, unsigned long long mask, int iters)
{
    double x0 = 1.0 + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const double a = 1.0000001, b = 1e-9;
    unsigned long long save;
    for (int i = 0; i < iters; ++i) {
        // (EXEC is narrowed and restored inside ONE asm block: nothing the compiler schedules ever runs under the narrowed mask)
#define R(x) "v_fma_f64 %" #x ", %" #x ", %9, %10\n\t"
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %11\n\t"
                     R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8)
                     R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8)
                     "s_mov_b64 exec, %0"
                     : "=&s"(save), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                     : "v"(a), "v"(b), "s"(mask));
#undef R
    }
    sink[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, wgs = cus * 4 * 8, iters = 20000;   // 8 single-wave workgroups per SIMD
    double *sink;
    CHECK(hipMalloc(&sink, (size_t)wgs * 64 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct { const char *name; unsigned long long mask; } cases[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-47", 0xffffffffffffull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15 (one quarter)", 0xffffull},
        {"lanes 16-31", 0xffff0000ull}, {"lanes 48-63", 0xffff000000000000ull}, {"4 lanes, one per quarter", 0x0001000100010001ull},
        {"4 lanes in one quarter", 0xfull}, {"1 lane", 1ull}, {"even lanes", 0x5555555555555555ull}};
    for (int pass = 0; pass < 2; ++pass)
        for (auto &c : cases) {
            fma_stream<<<wgs, 64>>>(sink, c.mask, 2000);   // warm
            CHECK(hipEventRecord(e0));
            fma_stream<<<wgs, 64>>>(sink, c.mask, iters);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_simd = 8.0 * iters * 32.0;
            if (pass) printf("%-28s %8.3f ms  %6.3f ns per wave-instruction per SIMD  (= %.2f cycles at 2.4 GHz)\n", c.name, ms, ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
        }
    return 0;
}

// anyorder.hip -- does hipExtAnyOrderLaunch (AQL packet without the barrier bit) let kernel B start while kernel A, launched
// before it on the SAME stream, is still running on gfx950?  (round 6: the dispatch-order pre-pass as an any-order launch,
// MBK_OPT_PREPASS_OVERLAP = 3.)   hipcc --offload-arch=gfx950 -O3 -o /tmp/anyorder profiles/microbench/anyorder.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>

__global__ void spin(unsigned long long ticks, unsigned long long *out)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}

int main()
{
    unsigned long long *h;
    hipHostMalloc((void **)&h, 64, hipHostMallocDefault);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 20000ull, h);          // A: 200 us at 100 MHz
            if (mode) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, 1000ull, h + 2);
            else hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000ull, h + 2);   // B: 10 us
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1000ull, h + 4);        // C: ordinary, must follow both
            hipStreamSynchronize(s);
            printf("%s B: A %.1f us long; B starts %+.1f us relative to A's END; C starts %+.1f us after max(A, B) end\n",
                   mode ? "any-order" : "ordinary ", (h[1] - h[0]) / 100.0, ((double)h[2] - (double)h[1]) / 100.0,
                   ((double)h[4] - (double)(h[1] > h[3] ? h[1] : h[3])) / 100.0);
        }
    return 0;
}

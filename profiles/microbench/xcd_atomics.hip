// xcd_atomics.hip -- round 6: how fast is an atomic with return when the counter is PRIVATE to one XCD?
// Round 4 measured ~25 ns per device-scope atomic with return, serialised chip-wide whatever the address (units_pool_ab.txt):
// on a multi-XCD chip a device-scope atomic bypasses the XCD's L2 and is resolved on the memory side.  A counter that only the
// waves of ONE XCD ever touch needs no more than that XCD's L2 -- which is where an atomic of workgroup scope is executed
// (all CUs of an XCD share the L2; the vector L1 executes no atomics).  This probe: N single-wave workgroups, each does one
// fetch-add of its lane count on counter[XCC_ID] (a) device scope, (b) workgroup scope, then the host checks the sums and, for
// (b), that the returned bases of each XCD are a permutation of the multiples (no two waves got the same base).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_atomics profiles/microbench/xcd_atomics.hip && /tmp/xcd_atomics
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int kScope>
__global__ __launch_bounds__(64) void bump(uint32_t *ctr, uint32_t *base_out, uint32_t *xcc_out, uint32_t spin)
{
    // a little arithmetic first so that the atomics of a launch are spread in time like a real kernel's
    double z = 0.1 + 1e-9 * threadIdx.x;
    for (uint32_t i = 0; i < spin; ++i) z = z * z + 0.25;
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu;    // HW_REG_XCC_ID, bits 3:0
    uint32_t base = 0;
    if (threadIdx.x == 0) {
        if (kScope == 0) base = __hip_atomic_fetch_add(&ctr[xcc * 32u], 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else base = __hip_atomic_fetch_add(&ctr[xcc * 32u], 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base_out[blockIdx.x] = base + (z > 1e300 ? 1u : 0u);
        xcc_out[blockIdx.x] = xcc;
    }
}

int main()
{
    const uint32_t N = 1u << 20;
    uint32_t *ctr, *base, *xcc;
    CHECK(hipMalloc(&ctr, 8 * 32 * 4));
    CHECK(hipMalloc(&base, N * 4));
    CHECK(hipMalloc(&xcc, N * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int scope = 0; scope < 2; ++scope)
        for (uint32_t spin : {0u, 200u, 2000u}) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipMemset(ctr, 0, 8 * 32 * 4));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                if (scope == 0) bump<0><<<N, 64>>>(ctr, base, xcc, spin); else bump<1><<<N, 64>>>(ctr, base, xcc, spin);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = std::min(best, ms);
            }
            std::vector<uint32_t> hc(8 * 32), hb(N), hx(N);
            CHECK(hipMemcpy(hc.data(), ctr, 8 * 32 * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hb.data(), base, N * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hx.data(), xcc, N * 4, hipMemcpyDeviceToHost));
            uint64_t sum = 0;
            uint32_t per[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            bool ok = true;
            std::vector<std::vector<uint32_t>> bases(8);
            for (uint32_t i = 0; i < N; ++i) { if (hx[i] > 7) { ok = false; continue; } per[hx[i]]++; bases[hx[i]].push_back(hb[i]); }
            for (int x = 0; x < 8; ++x) {
                sum += hc[x * 32];
                ok = ok && hc[x * 32] == 7u * per[x];
                std::sort(bases[x].begin(), bases[x].end());
                for (size_t j = 0; j < bases[x].size(); ++j) ok = ok && bases[x][j] == 7u * (uint32_t)j;
            }
            printf("%s scope, spin %4u: %u workgroups, one atomic with return each on the counter of their own XCD: %.3f ms (%.1f ns per atomic, chip-wide); "
                   "sums %s, bases %s; workgroups per XCD %u %u %u %u %u %u %u %u\n", scope ? "workgroup" : "device   ", spin, N, best, best * 1e6 / N,
                   sum == 7ull * N ? "exact" : "WRONG", ok ? "a permutation (atomic)" : "NOT ATOMIC", per[0], per[1], per[2], per[3], per[4], per[5], per[6], per[7]);
        }
    return 0;
}

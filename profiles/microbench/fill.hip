// fill.hip -- the yardstick for the HBM-bound launches: how fast can THIS box write a DataChunk-sized
// output at all, and what does the product's 8x8-block store order cost against row-contiguous 16-byte stores?
// No arithmetic: every variant stores a constant derived from the lane number.
//   64 MiB int32 (4096 x 4096 counts) and 16 MiB uint8 (the quantised tile):
//     memset        hipMemsetAsync (the runtime's own fill kernel)
//     block8x8      one 8x8 block per wave trip, lane (lx, ly) stores one element: 8 row segments of 32 B (int32) / 8 B (uint8)
//                   per wave instruction, blocks in image order
//     block8x8x     the same, XCD-aware block-column order of kernel "scan" (a 128-byte line is completed inside one XCD)
//     row_x1        64 consecutive elements per wave instruction (one 256 B / 64 B run)
//     row_x4        dwordx4: every lane 16 B, a wave instruction writes 1 KiB contiguous
//   each as a persistent grid (G = 256 CUs x 32 single-wave workgroups striding, the product's light pass) and as one
//   workgroup per 64-element-instruction's worth of work ("flat").
// Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fill profiles/microbench/fill.hip && /tmp/fill
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum Mode { BLOCK, BLOCKX, ROW1, ROW4 };

// T = int32_t or uint8_t; the image is 4096 x 4096 elements of T.  Work item w of `nwork`:
//   BLOCK / BLOCKX: 8x8 block w (512 blocks per block row); ROW1: 64-element run w; ROW4: run w of 64 x 16 B.
template <typename T, int MODE>
__global__ __launch_bounds__(64) void fill_kernel(T *out, uint32_t nwork, uint32_t stride)
{
    const uint32_t lane = threadIdx.x;
    for (uint32_t w = blockIdx.x; w < nwork; w += stride) {
        if (MODE == BLOCK || MODE == BLOCKX) {
            uint32_t by = w >> 9, bx = w & 511u;
            if (MODE == BLOCKX) {
                const uint32_t a = bx >> 3, c = bx & 7u;
                bx = ((a >> 2) << 5) | (c << 2) | (a & 3u);
            }
            out[(size_t)(by * 8u + (lane >> 3)) * 4096u + bx * 8u + (lane & 7u)] = (T)lane;
        } else if (MODE == ROW1) {
            out[(size_t)w * 64u + lane] = (T)lane;
        } else {
            uint4 v = {lane, lane, lane, lane};
            reinterpret_cast<uint4 *>(out)[(size_t)w * 64u + lane] = v;
        }
    }
}

template <typename T, int MODE>
static double time_fill(T *d, uint32_t nwork, uint32_t grid, int reps, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((fill_kernel<T, MODE>), dim3(grid), dim3(64), 0, s, d, nwork, grid);
    CHECK(hipStreamSynchronize(s));
    std::vector<float> ms;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((fill_kernel<T, MODE>), dim3(grid), dim3(64), 0, s, d, nwork, grid);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float t;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t / 50.f);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2] * 1e3;   // us, median of `reps` averages over 50 back-to-back launches
}

template <typename T>
static void run(const char *what, hipStream_t s, hipEvent_t e0, hipEvent_t e1)
{
    const size_t n = 4096ull * 4096ull, bytes = n * sizeof(T);
    T *d = nullptr;
    CHECK(hipMalloc((void **)&d, bytes));
    const uint32_t G = 256u * 32u;
    printf("== %s: %zu bytes\n", what, bytes);
    {   // memset
        for (int i = 0; i < 20; ++i) CHECK(hipMemsetAsync(d, 1, bytes, s));
        CHECK(hipStreamSynchronize(s));
        std::vector<float> ms;
        for (int r = 0; r < 7; ++r) {
            CHECK(hipEventRecord(e0, s));
            for (int i = 0; i < 50; ++i) CHECK(hipMemsetAsync(d, 1, bytes, s));
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float t;
            CHECK(hipEventElapsedTime(&t, e0, e1));
            ms.push_back(t / 50.f);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-28s %8.2f us  %6.2f TB/s\n", "memset", ms[3] * 1e3, bytes / (ms[3] * 1e-3) / 1e12);
    }
    const uint32_t nblocks = (uint32_t)(n / 64u), nrow4 = (uint32_t)(bytes / 1024u);
    struct { const char *name; double us; } rows[] = {
        {"block8x8   persistent", time_fill<T, BLOCK>(d, nblocks, G, 7, s, e0, e1)},
        {"block8x8   flat", time_fill<T, BLOCK>(d, nblocks, nblocks, 7, s, e0, e1)},
        {"block8x8x  persistent", time_fill<T, BLOCKX>(d, nblocks, G, 7, s, e0, e1)},
        {"block8x8x  flat", time_fill<T, BLOCKX>(d, nblocks, nblocks, 7, s, e0, e1)},
        {"row_x1     persistent", time_fill<T, ROW1>(d, nblocks, G, 7, s, e0, e1)},
        {"row_x1     flat", time_fill<T, ROW1>(d, nblocks, nblocks, 7, s, e0, e1)},
        {"row_x4     persistent", time_fill<T, ROW4>(d, nrow4, G, 7, s, e0, e1)},
        {"row_x4     flat", time_fill<T, ROW4>(d, nrow4, nrow4, 7, s, e0, e1)},
    };
    for (auto &r : rows) printf("%-28s %8.2f us  %6.2f TB/s\n", r.name, r.us, bytes / (r.us * 1e-6) / 1e12);
    CHECK(hipFree(d));
}

int main()
{
    hipStream_t s;
    hipEvent_t e0, e1;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // wake the clock
    {
        int32_t *d;
        CHECK(hipMalloc((void **)&d, 64u << 20));
        for (int i = 0; i < 4000; ++i) hipLaunchKernelGGL((fill_kernel<int32_t, ROW4>), dim3(8192), dim3(64), 0, s, d, 65536u, 8192u);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipFree(d));
    }
    run<int32_t>("int32 counts tile, 4096 x 4096", s, e0, e1);
    run<uint8_t>("uint8 byte tile, 4096 x 4096", s, e0, e1);
    return 0;
}

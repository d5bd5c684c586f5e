// light_path.hip -- what bounds pass 1 of the "scan" kernel on an all-exterior tile?  Splits the light path
// (distributedmandelbrot_amd/csrc/mbk_loops.inc: escape_light_block) into its parts on the DataChunk (4,0,0)
// geometry, 4096x4096 int32 counts, one 8x8 block per wave trip, persistent grid like the product:
//   compute   the arithmetic only (coordinates + branch-free prologue), result kept in registers
//   store8x8  the stores only, same 8x8 pattern (8 row segments of 32 B per wave instruction)
//   store8x8x same, XCD-aware column order (a 128-byte line is completed by one XCD)
//   store64x1 the stores only, 64 consecutive pixels per wave instruction (one 256-byte run)
//   both      compute + store8x8 (round 2's light path)
//   both_x    compute + store8x8x (= the product's light path since the XCD-aware order)
//   strip_c / strip   the same asm on 64x1 rows (lane = column, 8 rows per 64x8 region): arithmetic only / with stores
// Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I. -o /tmp/light_path profiles/microbench/light_path.hip && /tmp/light_path
#include "distributedmandelbrot_amd/csrc/mbk_refill.h"  // mbk_kernels.h + the wave-uniform helpers

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// The arithmetic of the light path alone (modes compute / strip_c): four branch-free steps of the reference loop, an
// escaped lane keeps iterating, cnt = 1 + number of steps it stayed inside (round 2's light path; the product now
// narrows EXEC instead -- mbk_loops.inc: escape_light_run -- at the same instruction count per step).
#define LP_STEP                                  \
    "v_add_f64 %[t], %[a], -%[b]\n"              \
    "v_mul_f64 %[p], %[zr], %[zi]\n"             \
    "v_add_f64 %[zr], %[t], %[cr]\n"             \
    "v_fma_f64 %[zi], %[p], 2.0, %[ci]\n"        \
    "v_mul_f64 %[a], %[zr], %[zr]\n"             \
    "v_mul_f64 %[b], %[zi], %[zi]\n"             \
    "v_add_f64 %[m], %[a], %[b]\n"               \
    "v_cmp_gt_f64 vcc, 4.0, %[m]\n"              \
    "v_addc_co_u32_e64 %[cnt], %[tmp], %[cnt], 0, vcc\n" \
    "s_cbranch_vccz .Lprodone_%=\n"
__device__ __forceinline__ void prologue4(double cr, double ci, double &zr, double &zi, double &a, double &b, int32_t &cnt)
{
    double t, p, m;
    unsigned long long tmp;
    cnt = 1;
    asm volatile(LP_STEP LP_STEP LP_STEP LP_STEP ".Lprodone_%=:\n"
                 : [zr] "+&v"(zr), [zi] "+&v"(zi), [a] "+&v"(a), [b] "+&v"(b), [cnt] "+&v"(cnt), [m] "=&v"(m),
                   [t] "=&v"(t), [p] "=&v"(p), [tmp] "=&s"(tmp)
                 : [cr] "v"(cr), [ci] "v"(ci)
                 : "vcc");
}

enum Mode { COMPUTE, STORE8, STORE8X, STORE64, BOTH, BOTHX, STRIP, STRIPC };

template <int MODE>
__global__ __launch_bounds__(64) void light_kernel(int32_t *out, double start, double step, uint32_t stride_by, uint32_t nby,
                                                   int32_t *sink)
{
    const uint32_t lane = threadIdx.x, lx = lane & 7u, ly = lane >> 3;
    const uint32_t blocks_x = 512u, pitch = 4096u;
    uint32_t by = blockIdx.x / blocks_x, bx = blockIdx.x - by * blocks_x;
    if (MODE == STORE8X || MODE == BOTHX) {
        const uint32_t a = bx >> 3, c = bx & 7u;
        bx = ((a >> 2) << 5) | (c << 2) | (a & 3u);
    }
    const double cr = (double)(bx * 8u + lx) * step + start, a0 = cr * cr;
    const uint32_t lane_elem = ly * pitch + lx;
    int32_t acc = 0;
    if (MODE == STORE64) {   // 64x1 strips: wave w writes 256-byte runs; same number of store instructions
        uint32_t run = blockIdx.x;                                    // 64 runs per row, 4096 rows
        for (; run < 64u * 4096u; run += gridDim.x * 0u + 7168u) out[(size_t)run * 64u + lane] = (int32_t)lane;
        return;
    }
    if (MODE == STRIP || MODE == STRIPC) {
        // 64x1 strips: lane = column, one image row per trip (ci wave-uniform), 8 rows = one 64x8 region; wave w takes
        // regions w, w + W, ... (W a multiple of 64: a wave keeps its 64 columns).  Same asm, same arithmetic per pixel;
        // every store instruction writes one 256-byte run.
        const uint32_t rx = blockIdx.x & 63u;
        const double crs = (double)(rx * 64u + lane) * step + start, a0s = crs * crs;
        for (uint32_t band = blockIdx.x >> 6; band < 512u; band += 7168u / 64u) {
            for (uint32_t r = 0; r < 8u; ++r) {
                const uint32_t row = band * 8u + r;
                double ci, zr, zi, a, b;
                int32_t cnt;
                int32_t *rb = out + (size_t)row * pitch + rx * 64u;
                int32_t *cb = reinterpret_cast<int32_t *>(mbk::uniform_u64(reinterpret_cast<unsigned long long>(rb)));
                if (MODE == STRIP) {
                    uint32_t rowv = row, off = lane * 4u, n = 1u;
                    mbk::escape_light_run<true, false>(crs, a0s, rowv, 1u, step, start, cnt, cb, nullptr, off, 0u, 0u, n);
                } else {
                    ci = (double)row * step + start;
                    zr = crs; zi = ci; a = a0s; b = ci * ci;
                    prologue4(crs, ci, zr, zi, a, b, cnt);
                }
                acc += cnt;
            }
        }
        if (acc == 0x7fffffff) *sink = acc;
        return;
    }
    if (MODE == BOTH || MODE == BOTHX) {   // the product's form: ONE asm loop over the wave's whole run of blocks
        int32_t *base = out + (size_t)(by * 8u) * pitch + bx * 8u;
        int32_t *cb = reinterpret_cast<int32_t *>(mbk::uniform_u64(reinterpret_cast<unsigned long long>(base)));
        uint32_t row = by * 8u + ly, off = lane_elem * 4u, n = mbk::uniform_u32((nby - by + stride_by - 1u) / stride_by);
        int32_t cnt;
        uint32_t more = n;
        while (more != 0u) {   // (an unfinished block is skipped here; loop shape as in tile_light_kernel)
            more = mbk::escape_light_run<true, false>(cr, a0, row, stride_by * 8u, step, start, cnt, cb, nullptr, off, stride_by * 8u * pitch * 4u, 0u, n);
            row += stride_by * 8u; off += stride_by * 8u * pitch * 4u; --n;
            more = more != 0u ? n : 0u;
        }
        return;
    }
    for (; by < nby; by += stride_by) {
        int32_t *base = out + (size_t)(by * 8u) * pitch + bx * 8u;
        if (MODE == STORE8 || MODE == STORE8X) {
            base[lane_elem] = (int32_t)by;
        } else {
            double ci, zr, zi, a, b;
            int32_t cnt;
            int32_t *cb = reinterpret_cast<int32_t *>(mbk::uniform_u64(reinterpret_cast<unsigned long long>(base)));
            {   // arithmetic only: the plain prologue on the same coordinates
                ci = (double)(by * 8u + ly) * step + start;
                zr = cr; zi = ci; a = a0; b = ci * ci;
                prologue4(cr, ci, zr, zi, a, b, cnt);
            }
            acc += cnt;
        }
    }
    if (MODE == COMPUTE && acc == 0x7fffffff) *sink = acc;
}

template <int MODE>
static void run(const char *name, int32_t *d_out, int32_t *d_sink)
{
    const uint32_t grid = 7168u, stride_by = grid / 512u, nby = 512u;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 200; ++i) light_kernel<MODE><<<grid, 64>>>(d_out, -2.0, 1.0 / 4095.0, stride_by, nby, d_sink);   // clock ramp
    CHECK(hipDeviceSynchronize());
    const int reps = 400;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) light_kernel<MODE><<<grid, 64>>>(d_out, -2.0, 1.0 / 4095.0, stride_by, nby, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-10s %7.2f us per 4096^2 tile (262144 blocks)   %6.2f TB/s of int32 output\n", name, ms / reps * 1e3,
           MODE == COMPUTE ? 0.0 : 67108864.0 / (ms / reps * 1e-3) / 1e12);
}

int main()
{
    int32_t *d_out, *d_sink;
    CHECK(hipMalloc(&d_out, 4096ull * 4096 * 4)); CHECK(hipMalloc(&d_sink, 4));
    run<COMPUTE>("compute", d_out, d_sink);
    run<STORE8>("store8x8", d_out, d_sink);
    run<STORE8X>("store8x8x", d_out, d_sink);
    run<STORE64>("store64x1", d_out, d_sink);
    run<BOTH>("both", d_out, d_sink);
    run<BOTHX>("both_x", d_out, d_sink);      // compute + store8x8x: what the product's pass 1 does per block
    run<STRIPC>("strip_c", d_out, d_sink);    // 64x1 rows: the arithmetic only
    run<STRIP>("strip", d_out, d_sink);       // 64x1 rows: compute + one 256-byte run per store instruction
    return 0;
}

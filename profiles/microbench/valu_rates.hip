// valu_rates.hip -- gfx950 VALU issue-rate / latency microbenchmark that calibrates the roofline of
// the escape-time kernel (fp64 mul/add/fma, f64 and u32 compares, 32-bit VALU helpers), and the
// Mandelbrot loop body itself with no exit test.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates profiles/microbench/valu_rates.hip && /tmp/valu_rates
// Cycles come from s_memtime (shader clock), so DVFS does not distort them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)



enum Kind { FMA_IND, MUL_IND, ADD_IND, FMA_DEP, MUL_DEP, ADD_DEP, CMP_F64, CMP_U32, ADD_U32, MOV_B32, FMA_F32, BODY, BODY2, BODY_CMP,
            PK_FMA_F32, PK_MUL_F32, PK_ADD_F32, MUL_F32, ADD_F32, FMA_CONST, SQR_F64, BODY_PK };
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long *cycles, unsigned long long *wall, double *sink, double seed, int ITERS)
{
    double x0 = seed + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    double a = 1.0000001, b = 1e-9;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, fa = 1.0001f, fb = 1e-5f;
    // mandelbrot state (c inside the set so nothing overflows): two independent pixels
    double cr = -0.1 + 1e-6 * threadIdx.x, ci = 0.2, zr = cr, zi = ci, aa = zr * zr, bb = zi * zi, t, p, m;
    double cr2 = -0.2 + 1e-6 * threadIdx.x, ci2 = 0.1, zr2 = cr2, zi2 = ci2, aa2 = zr2 * zr2, bb2 = zi2 * zi2, t2, p2, m2;
    // packed fp32: two values per lane in a 64-bit register pair (v_pk_*_f32)
    float2_ q0 = {f0, f1}, q1 = {f1, f2}, q2 = {f2, f3}, q3 = {f3, f0}, q4 = {f0 + 4, f1}, q5 = {f1 + 4, f2}, q6 = {f2 + 4, f3},
            q7 = {f3 + 4, f0}, qa = {fa, fa}, qb = {fb, fb};
    // packed mandelbrot state: two pixels per lane
    float2_ pcr = {-0.1f + 1e-4f * threadIdx.x, -0.2f + 1e-4f * threadIdx.x}, pci = {0.2f, 0.1f}, pzr = pcr, pzi = pci,
            paa = pzr * pzr, pbb = pzi * pzi, pt, pp, ptwo = {2.0f, 2.0f};
    unsigned long long w0 = wall_clock64();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; ++i) {
        if (KIND == FMA_IND) {
#define R(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7) R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7)
#undef R
        } else if (KIND == MUL_IND) {
#define R(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a));
            R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7) R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7)
#undef R
        } else if (KIND == ADD_IND) {
#define R(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
            R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7) R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7)
#undef R
        } else if (KIND == FMA_DEP) {
#define R(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0)
#undef R
        } else if (KIND == MUL_DEP) {
#define R(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a));
            R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0)
#undef R
        } else if (KIND == ADD_DEP) {
#define R(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
            R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0) R(x0)
#undef R
        } else if (KIND == CMP_F64) {
#define R(x) asm volatile("v_cmp_ge_f64 vcc, %0, %1" : : "v"(x), "v"(a) : "vcc");
            R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7) R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7)
#undef R
        } else if (KIND == CMP_U32) {
#define R(x) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x), "v"(u3) : "vcc");
            R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0)
#undef R
        } else if (KIND == ADD_U32) {
#define R(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(u3));
            R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0)
#undef R
        } else if (KIND == MOV_B32) {
#define R(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(u3));
            R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0) R(u1) R(u2) R(u0)
#undef R
        } else if (KIND == FMA_F32) {
#define R(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(fa), "v"(fb));
            R(f0) R(f1) R(f2) R(f3) R(f0) R(f1) R(f2) R(f3) R(f0) R(f1) R(f2) R(f3) R(f0) R(f1) R(f2) R(f3)
#undef R
        } else if (KIND == PK_FMA_F32) {
#define R(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(qa), "v"(qb));
            R(q0) R(q1) R(q2) R(q3) R(q4) R(q5) R(q6) R(q7) R(q0) R(q1) R(q2) R(q3) R(q4) R(q5) R(q6) R(q7)
#undef R
        } else if (KIND == PK_MUL_F32) {
#define R(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(qa));
            R(q0) R(q1) R(q2) R(q3) R(q4) R(q5) R(q6) R(q7) R(q0) R(q1) R(q2) R(q3) R(q4) R(q5) R(q6) R(q7)
#undef R
        } else if (KIND == PK_ADD_F32) {
#define R(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(qb));
            R(q0) R(q1) R(q2) R(q3) R(q4) R(q5) R(q6) R(q7) R(q0) R(q1) R(q2) R(q3) R(q4) R(q5) R(q6) R(q7)
#undef R
        } else if (KIND == MUL_F32) {
#define R(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(fa));
            R(f0) R(f1) R(f2) R(f3) R(f0) R(f1) R(f2) R(f3) R(f0) R(f1) R(f2) R(f3) R(f0) R(f1) R(f2) R(f3)
#undef R
        } else if (KIND == ADD_F32) {
#define R(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(fb));
            R(f0) R(f1) R(f2) R(f3) R(f0) R(f1) R(f2) R(f3) R(f0) R(f1) R(f2) R(f3) R(f0) R(f1) R(f2) R(f3)
#undef R
        } else if (KIND == FMA_CONST) {
#define R(x) asm volatile("v_fma_f64 %0, %0, 2.0, %1" : "+v"(x) : "v"(b));
            R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7) R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7)
#undef R
        } else if (KIND == SQR_F64) {
#define R(x) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(x));
            R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7) R(x0) R(x1) R(x2) R(x3) R(x4) R(x5) R(x6) R(x7)
#undef R
        } else if (KIND == BODY_PK) {
            // two packed-fp32 steps: 6 packed ops per step = 2 pixels per lane
#define IT                                                                                       \
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(pt) : "v"(paa), "v"(pbb));   \
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pp) : "v"(pzr), "v"(pzi));                \
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(pzr) : "v"(pt), "v"(pcr));                \
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(pzi) : "v"(pp), "v"(ptwo), "v"(pci)); \
            asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(paa) : "v"(pzr));                         \
            asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(pbb) : "v"(pzi));
            IT IT
#undef IT
        } else if (KIND == BODY || KIND == BODY_CMP) {
            // two iterations of one pixel per loop trip (7 fp64 ops each)
#define IT                                                                                       \
            asm volatile("v_add_f64 %0, %1, -%2" : "=v"(t) : "v"(aa), "v"(bb));                     \
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p) : "v"(zr), "v"(zi));                      \
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(zr) : "v"(t), "v"(cr));                      \
            asm volatile("v_fma_f64 %0, %1, 2.0, %2" : "=v"(zi) : "v"(p), "v"(ci));                 \
            asm volatile("v_mul_f64 %0, %1, %1" : "=v"(aa) : "v"(zr));                              \
            asm volatile("v_mul_f64 %0, %1, %1" : "=v"(bb) : "v"(zi));                              \
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(m) : "v"(aa), "v"(bb));                      \
            if (KIND == BODY_CMP) asm volatile("v_cmp_gt_u32 vcc, 0x40100000, %0" : : "v"((unsigned)(__double_as_longlong(m) >> 32)) : "vcc");
            IT IT
#undef IT
        } else if (KIND == BODY2) {
            // two independent pixels per lane interleaved (ILP 2)
            asm volatile("v_add_f64 %0, %1, -%2" : "=v"(t) : "v"(aa), "v"(bb));
            asm volatile("v_add_f64 %0, %1, -%2" : "=v"(t2) : "v"(aa2), "v"(bb2));
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p) : "v"(zr), "v"(zi));
            asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p2) : "v"(zr2), "v"(zi2));
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(zr) : "v"(t), "v"(cr));
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(zr2) : "v"(t2), "v"(cr2));
            asm volatile("v_fma_f64 %0, %1, 2.0, %2" : "=v"(zi) : "v"(p), "v"(ci));
            asm volatile("v_fma_f64 %0, %1, 2.0, %2" : "=v"(zi2) : "v"(p2), "v"(ci2));
            asm volatile("v_mul_f64 %0, %1, %1" : "=v"(aa) : "v"(zr));
            asm volatile("v_mul_f64 %0, %1, %1" : "=v"(aa2) : "v"(zr2));
            asm volatile("v_mul_f64 %0, %1, %1" : "=v"(bb) : "v"(zi));
            asm volatile("v_mul_f64 %0, %1, %1" : "=v"(bb2) : "v"(zi2));
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(m) : "v"(aa), "v"(bb));
            asm volatile("v_add_f64 %0, %1, %2" : "=v"(m2) : "v"(aa2), "v"(bb2));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    unsigned long long w1 = wall_clock64();
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    if ((threadIdx.x & 63) == 0) { cycles[gid >> 6] = t1 - t0; wall[gid >> 6] = w1 - w0; }
    double r = 0;
    if (KIND <= ADD_DEP || KIND == CMP_F64) r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    else if (KIND == CMP_U32 || KIND == ADD_U32 || KIND == MOV_B32) r = u0 + u1 + u2;
    else if (KIND == FMA_F32 || KIND == MUL_F32 || KIND == ADD_F32) r = f0 + f1 + f2 + f3;
    else if (KIND == PK_FMA_F32 || KIND == PK_MUL_F32 || KIND == PK_ADD_F32) { float2_ qs = q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7; r = qs.x + qs.y; }
    else if (KIND == BODY_PK) r = pzr.x + pzr.y + pzi.x + pzi.y + paa.x + pbb.y;
    else if (KIND == FMA_CONST || KIND == SQR_F64) r = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    else r = zr + zi + zr2 + zi2 + m + m2;
    sink[gid] = r;
}

static int instr_per_iter(int kind) { return kind == BODY ? 14 : kind == BODY_CMP ? 16 : kind == BODY2 ? 14 : kind == BODY_PK ? 12 : 16; }

template <int KIND>
void run(const char *name, int cus)
{
    int occ = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, rate_kernel<KIND>, 256, 0));
    for (int waves_per_simd : {1, 2, 3, 4, 6, 8}) {
        const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
        const int nwaves = blocks * 4;
        unsigned long long *d_cycles, *d_wall; double *d_sink;
        CHECK(hipMalloc(&d_cycles, nwaves * sizeof(unsigned long long)));
        CHECK(hipMalloc(&d_wall, nwaves * sizeof(unsigned long long)));
        CHECK(hipMalloc(&d_sink, (size_t)blocks * 256 * sizeof(double)));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float ms[2]; double med[2], wmed[2];
        const int iters[2] = {2000, 6000};
        for (int rep = 0; rep < 2; ++rep) {
            rate_kernel<KIND><<<blocks, 256>>>(d_cycles, d_wall, d_sink, 0.5, iters[rep]);  // warm-up
            CHECK(hipEventRecord(e0));
            rate_kernel<KIND><<<blocks, 256>>>(d_cycles, d_wall, d_sink, 0.5, iters[rep]);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms[rep], e0, e1));
            std::vector<unsigned long long> h(nwaves), hw(nwaves);
            CHECK(hipMemcpy(h.data(), d_cycles, nwaves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(hw.data(), d_wall, nwaves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end()); std::sort(hw.begin(), hw.end());
            med[rep] = (double)h[nwaves / 2]; wmed[rep] = (double)hw[nwaves / 2];
        }
        const double ipi = instr_per_iter(KIND);
        const double d_instr = (iters[1] - iters[0]) * ipi;            // extra instructions per wave
        const double ns_per_simd_instr = (ms[1] - ms[0]) * 1e6 / (d_instr * waves_per_simd);
        const double ctr_per_wave_instr = (med[1] - med[0]) / d_instr;  // s_memtime ticks
        const double wall_ns_per_wave_instr = (wmed[1] - wmed[0]) * 10.0 / d_instr;  // wall_clock64 = 100 MHz
        const double ctr_mhz = (med[1] - med[0]) / ((wmed[1] - wmed[0]) * 10.0) * 1e3;
        printf("%-10s occ %d blk/CU  waves/SIMD=%d | slope: %.3f ns per SIMD-instr (kernel time) | per wave: %.2f ticks/instr = %.2f ns/instr | s_memtime rate %.0f MHz | => SIMD cycles/instr @2.4GHz-equiv %.2f\n",
               name, occ, waves_per_simd, ns_per_simd_instr, ctr_per_wave_instr, wall_ns_per_wave_instr, ctr_mhz, ns_per_simd_instr * 2.4);
        CHECK(hipFree(d_cycles)); CHECK(hipFree(d_wall)); CHECK(hipFree(d_sink));
    }
}

int main(int argc, char **argv)
{
    const bool only_new = argc > 1;   // any argument: only the kinds added in round 2
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s arch %s CUs %d clock %d MHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const int cus = prop.multiProcessorCount;
    run<PK_FMA_F32>("pk_fma_f32", cus);
    run<PK_MUL_F32>("pk_mul_f32", cus);
    run<PK_ADD_F32>("pk_add_f32", cus);
    run<FMA_F32>("fma_f32", cus);
    run<MUL_F32>("mul_f32", cus);
    run<ADD_F32>("add_f32", cus);
    run<BODY_PK>("body_pk", cus);
    run<FMA_CONST>("fma_f64c", cus);
    run<SQR_F64>("sqr_f64", cus);
    if (only_new) return 0;
    run<FMA_IND>("fma_f64", cus);
    run<MUL_IND>("mul_f64", cus);
    run<ADD_IND>("add_f64", cus);
    run<FMA_DEP>("fma_f64dep", cus);
    run<MUL_DEP>("mul_f64dep", cus);
    run<ADD_DEP>("add_f64dep", cus);
    run<CMP_F64>("cmp_f64", cus);
    run<CMP_U32>("cmp_u32", cus);
    run<ADD_U32>("add_u32", cus);
    run<MOV_B32>("mov_b32", cus);
    run<FMA_F32>("fma_f32", cus);
    run<BODY>("body", cus);
    run<BODY_CMP>("body+cmp", cus);
    run<BODY2>("body2px", cus);
    return 0;
}

// mfma_f64_coissue.hip -- can the fp64 matrix pipe of gfx950 take one of the escape loop's six per-step operations
// off the vector ALU, bit-exactly?
//
// Idea.  v_mfma_f64_4x4x4_4b_f64 computes, per 4x4 block, D[i][j] = sum_k A[i][k] * B[k][j] + C[i][j] with one f64 of
// A, B, C, D per lane.  With A = s * identity (a per-lane CONSTANT: s where i == k, 0 elsewhere) this is, lane by lane,
//     D = s * B + C      (one rounding: the other three products are exact zeros)
// i.e. v_fma_f64 D, s, B, C for s in {1, -1, 2} -- each of "t = a - b", "zr' = t + cr", "zi' = fma(2, p, ci)" of the
// Mandelbrot step -- executed by the matrix core while the SIMD's vector ALU issues another wave's instruction.
// Conditions the experiment has to establish:
//   A. the lane maps (which lane holds A[i][k], B[k][j], D[i][j]) and whether an "identity" pattern exists that leaves
//      every result in the lane its operand came from; what EXEC does to an MFMA;
//   B. bit-exactness of D against v_fma_f64 over random operands incl. subnormals, and the known limit: a non-finite
//      operand in ANOTHER lane of the same block column turns 0 * inf into NaN (so the trick needs finite states);
//   C. throughput: the loop body with 0, 1, 2 operations moved to the matrix pipe, 8 and 4 waves per SIMD, and whether the
//      software wait states LLVM prescribes (VALU write -> MFMA read 2, DMFMA 4x4x4 write -> VALU read 6) are needed.
// Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/mfma_f64 profiles/microbench/mfma_f64_coissue.hip && /tmp/mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---------------------------------------------------------------- A. lane maps
// out[p * 64 + lane]: D with a one-hot operand at lane p (1.0) and the other operand = lane + 1
__global__ void probe_map(double *outA, double *outB)
{
    const int lane = threadIdx.x;
    for (int p = 0; p < 64; ++p) {
        const double hot = lane == p ? 1.0 : 0.0, idx = (double)(lane + 1);
        outA[p * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(hot, idx, 0.0, 0, 0, 0);  // one-hot A, B = lane + 1
        outB[p * 64 + lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(idx, hot, 0.0, 0, 0, 0);  // A = lane + 1, one-hot B
    }
}

// EXEC: lanes >= 32 are switched off around the MFMA; D pre-set to a sentinel; an inf sits in B of lane `inf_lane`
__global__ void probe_exec(double *out, unsigned long long maskA, int inf_lane)
{
    const int lane = threadIdx.x;
    double a = (maskA >> lane) & 1ull ? 1.0 : 0.0;
    double b = lane == inf_lane ? __builtin_inf() : (double)(lane + 1);
    double c = 0.5, d = -777.0;
    if (lane < 32) {
        asm volatile("s_nop 4\n v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %3\n s_nop 7" : "+v"(d) : "v"(a), "v"(b), "v"(c));
    }
    out[lane] = d;
}

// ---------------------------------------------------------------- B. exactness
// d_mfma = pattern(s) x B + C against v_fma_f64(s, B, C); operand in A or in B according to `data_in_a`
__global__ void exact_kernel(const double *x, const double *c, double *d_mfma, double *d_valu, unsigned long long mask,
                             double s, int data_in_a, int n)
{
    const int lane = threadIdx.x & 63;
    const double pat = (mask >> lane) & 1ull ? s : 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double xv = x[i], cv = c[i];
        double r;
        if (data_in_a) r = __builtin_amdgcn_mfma_f64_4x4x4f64(xv, pat, cv, 0, 0, 0);
        else r = __builtin_amdgcn_mfma_f64_4x4x4f64(pat, xv, cv, 0, 0, 0);
        double v;
        asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(v) : "v"(s), "v"(xv), "v"(cv));
        d_mfma[i] = r;
        d_valu[i] = v;
    }
}

// ---------------------------------------------------------------- C. throughput
// One Mandelbrot step on (zr, zi, a = zr^2, b = zi^2); 16 steps per asm block.  P1/P2/PN: the per-lane patterns
// 1*I, 2*I, -1*I.  NOPx: wait states after an MFMA before the first VALU read of its result.
#define OPERANDS                                                                                         \
    : [zr] "+&v"(zr), [zi] "+&v"(zi), [a] "+&v"(a), [b] "+&v"(b), [t] "=&v"(t), [p] "=&v"(p)             \
    : [cr] "v"(cr), [ci] "v"(ci), [P1] "v"(p1), [P2] "v"(p2), [PN] "v"(pn)
// V0: the product loop's step, all on the vector ALU (6 fp64 issue slots)
#define STEP_V0                                   \
    "v_add_f64 %[t], %[a], -%[b]\n"               \
    "v_mul_f64 %[p], %[zr], %[zi]\n"              \
    "v_add_f64 %[zr], %[t], %[cr]\n"              \
    "v_fma_f64 %[zi], %[p], 2.0, %[ci]\n"         \
    "v_mul_f64 %[a], %[zr], %[zr]\n"              \
    "v_mul_f64 %[b], %[zi], %[zi]\n"
// V1: zi' = 2p + ci on the matrix pipe (5 VALU + 1 MFMA)
#define STEP_V1(NOP)                              \
    "v_mul_f64 %[p], %[zr], %[zi]\n"              \
    "v_add_f64 %[t], %[a], -%[b]\n"               \
    "v_add_f64 %[zr], %[t], %[cr]\n"              \
    "v_mfma_f64_4x4x4_4b_f64 %[zi], %[P2], %[p], %[ci]\n" \
    "v_mul_f64 %[a], %[zr], %[zr]\n"              \
    NOP                                           \
    "v_mul_f64 %[b], %[zi], %[zi]\n"
// V2: zr' = t + cr on the matrix pipe
#define STEP_V2(NOP)                              \
    "v_add_f64 %[t], %[a], -%[b]\n"               \
    "v_mul_f64 %[p], %[zr], %[zi]\n"              \
    "v_fma_f64 %[zi], %[p], 2.0, %[ci]\n"         \
    "v_mfma_f64_4x4x4_4b_f64 %[zr], %[P1], %[t], %[cr]\n" \
    "v_mul_f64 %[b], %[zi], %[zi]\n"              \
    NOP                                           \
    "v_mul_f64 %[a], %[zr], %[zr]\n"
// V3: both (4 VALU + 2 MFMA)
#define STEP_V3(NOP)                              \
    "v_mul_f64 %[p], %[zr], %[zi]\n"              \
    "v_add_f64 %[t], %[a], -%[b]\n"               \
    "s_nop 0\n"                                   \
    "v_mfma_f64_4x4x4_4b_f64 %[zi], %[P2], %[p], %[ci]\n" \
    "v_mfma_f64_4x4x4_4b_f64 %[zr], %[P1], %[t], %[cr]\n" \
    NOP                                           \
    "v_mul_f64 %[b], %[zi], %[zi]\n"              \
    "v_mul_f64 %[a], %[zr], %[zr]\n"
// V4: t = a - b on the matrix pipe (t = -1 * b + a)
#define STEP_V4(NOP)                              \
    "v_mul_f64 %[p], %[zr], %[zi]\n"              \
    "v_fma_f64 %[zi], %[p], 2.0, %[ci]\n"         \
    "v_mfma_f64_4x4x4_4b_f64 %[t], %[PN], %[b], %[a]\n" \
    "v_mul_f64 %[b], %[zi], %[zi]\n"              \
    NOP                                           \
    "v_add_f64 %[zr], %[t], %[cr]\n"              \
    "v_mul_f64 %[a], %[zr], %[zr]\n"
// V5: five VALU only (the fma dropped: the bound a perfect overlap could reach)
#define STEP_V5                                   \
    "v_add_f64 %[t], %[a], -%[b]\n"               \
    "v_mul_f64 %[p], %[zr], %[zi]\n"              \
    "v_add_f64 %[zr], %[t], %[cr]\n"              \
    "v_mul_f64 %[a], %[zr], %[zr]\n"              \
    "v_mul_f64 %[b], %[zi], %[zi]\n"
// V6: the matrix pipe alone, one dependent MFMA per step
#define STEP_V6(NOP)                              \
    "v_mfma_f64_4x4x4_4b_f64 %[zi], %[P1], %[zi], %[ci]\n" \
    NOP
// V7: the matrix pipe alone, two independent chains
#define STEP_V7(NOP)                              \
    "v_mfma_f64_4x4x4_4b_f64 %[zi], %[P1], %[zi], %[ci]\n" \
    "v_mfma_f64_4x4x4_4b_f64 %[zr], %[P1], %[zr], %[cr]\n" \
    NOP
#define X4(S) S S S S
#define X16(S) X4(S) X4(S) X4(S) X4(S)

enum { V0, V1_N4, V1_N0, V2_N4, V3_N5, V4_N4, V5, V6_N5, V7_N4, V1_N2, NVAR };
static const char *kNames[NVAR] = {"V0  6 VALU (the product's step)", "V1  zi'=2p+ci on MFMA, s_nop 4", "V1  same, no nop (is there an interlock?)",
                                   "V2  zr'=t+cr on MFMA, s_nop 4", "V3  both on MFMA (4 VALU + 2 MFMA)", "V4  t=a-b on MFMA, s_nop 4",
                                   "V5  5 VALU (fma dropped: overlap bound)", "V6  MFMA only, 1 dependent chain", "V7  MFMA only, 2 chains",
                                   "V1  same, s_nop 2"};

template <int VAR>
__global__ __launch_bounds__(256) void step_kernel(double *out, unsigned long long *ticks, unsigned long long mask, int trips)
{
    const int lane = threadIdx.x & 63;
    const double on = (mask >> lane) & 1ull ? 1.0 : 0.0;
    const double p1 = on, p2 = 2.0 * on, pn = -on;
    // a point of the main cardioid per lane: the orbit stays bounded (no inf/NaN, no subnormals)
    const double cr = -0.1 + 1e-6 * (threadIdx.x + 256 * (blockIdx.x & 63)), ci = 0.2 + 1e-7 * lane;
    double zr = cr, zi = ci, a = zr * zr, b = zi * zi, t, p;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < trips; ++i) {
        if (VAR == V0) asm volatile(X16(STEP_V0) OPERANDS);
        else if (VAR == V1_N4) asm volatile(X16(STEP_V1("s_nop 4\n")) OPERANDS);
        else if (VAR == V1_N2) asm volatile(X16(STEP_V1("s_nop 2\n")) OPERANDS);
        else if (VAR == V1_N0) asm volatile(X16(STEP_V1("")) OPERANDS);
        else if (VAR == V2_N4) asm volatile(X16(STEP_V2("s_nop 4\n")) OPERANDS);
        else if (VAR == V3_N5) asm volatile(X16(STEP_V3("s_nop 5\n")) OPERANDS);
        else if (VAR == V4_N4) asm volatile(X16(STEP_V4("s_nop 4\n")) OPERANDS);
        else if (VAR == V5) asm volatile(X16(STEP_V5) OPERANDS);
        else if (VAR == V6_N5) asm volatile(X16(STEP_V6("s_nop 5\n")) OPERANDS);
        else if (VAR == V7_N4) asm volatile(X16(STEP_V7("s_nop 4\n")) OPERANDS);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    out[2 * g] = zr;
    out[2 * g + 1] = zi;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int VAR>
static double run_variant(int wgs, int trips, double *d_out, unsigned long long *d_ticks, unsigned long long mask, double *tick_avg)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 6; ++r) {   // the first launches also ramp the clock
        CHECK(hipEventRecord(e0));
        step_kernel<VAR><<<wgs, 256>>>(d_out, d_ticks, mask, trips);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) best = std::min(best, ms);
    }
    std::vector<unsigned long long> tk(wgs);
    CHECK(hipMemcpy(tk.data(), d_ticks, wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double s = 0;
    for (auto v : tk) s += (double)v;
    *tick_avg = s / wgs;
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return best;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd()
{
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}
static double from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
static uint64_t to_bits(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
// classes of operands: 0 orbit-like (|x| < 4), 1 any finite exponent, 2 subnormal, 3 near-cancelling pairs
static double gen(int cls)
{
    const uint64_t r = rnd();
    const uint64_t sign = r & 0x8000000000000000ull, frac = r & 0x000FFFFFFFFFFFFFull;
    switch (cls) {
    case 0: return from_bits(sign | ((uint64_t)(1023 - (rnd() % 40) + 1) << 52) | frac);
    case 1: { uint64_t e = 1 + rnd() % 2045; if (e > 2040) e = 2040; return from_bits(sign | (e << 52) | frac); }
    case 2: return from_bits(sign | (rnd() % 3 == 0 ? 0 : frac));
    default: return from_bits(sign | ((uint64_t)1023 << 52) | frac);
    }
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int cus = prop.multiProcessorCount;

    // ---- A
    double *dA, *dB;
    CHECK(hipMalloc(&dA, 64 * 64 * 8));
    CHECK(hipMalloc(&dB, 64 * 64 * 8));
    probe_map<<<1, 64>>>(dA, dB);
    std::vector<double> hA(4096), hB(4096);
    CHECK(hipMemcpy(hA.data(), dA, 4096 * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hB.data(), dB, 4096 * 8, hipMemcpyDeviceToHost));
    // one-hot A at lane p: D[d] = B[q] -> q = value - 1.  Pattern lanes: those p that feed some d from q == d.
    unsigned long long maskA = 0, maskB = 0;
    bool okA = true, okB = true;
    int fedA[64] = {0}, fedB[64] = {0};
    for (int p = 0; p < 64; ++p) {
        int same = 0, other = 0, same2 = 0, other2 = 0;
        for (int d = 0; d < 64; ++d) {
            const double v = hA[p * 64 + d], w = hB[p * 64 + d];
            if (v != 0.0) { if ((int)v - 1 == d) { ++same; ++fedA[d]; } else ++other; }
            if (w != 0.0) { if ((int)w - 1 == d) { ++same2; ++fedB[d]; } else ++other2; }
        }
        if (same && other) okA = false;
        if (same2 && other2) okB = false;
        if (same) maskA |= 1ull << p;
        if (same2) maskB |= 1ull << p;
    }
    for (int d = 0; d < 64; ++d) { if (fedA[d] != 1) okA = false; if (fedB[d] != 1) okB = false; }
    printf("A. one-hot A at lane p -> D lanes fed (lane:value-1 = source lane of B):\n");
    for (int p = 0; p < 64; p += 21) {
        printf("   p=%2d:", p);
        for (int d = 0; d < 64; ++d) if (hA[p * 64 + d] != 0.0) printf(" D[%d]<-B[%d]", d, (int)hA[p * 64 + d] - 1);
        printf("\n");
    }
    for (int p = 0; p < 64; p += 21) {
        printf("   one-hot B p=%2d:", p);
        for (int d = 0; d < 64; ++d) if (hB[p * 64 + d] != 0.0) printf(" D[%d]<-A[%d]", d, (int)hB[p * 64 + d] - 1);
        printf("\n");
    }
    printf("   identity pattern in A (data in B, result in the data's own lane): %s, lanes mask 0x%016llx\n", okA ? "EXISTS" : "no", maskA);
    printf("   identity pattern in B (data in A):                                  %s, lanes mask 0x%016llx\n", okB ? "EXISTS" : "no", maskB);
    if (!okA && !okB) { printf("no usable pattern -- stop\n"); return 0; }
    const int data_in_a = okA ? 0 : 1;
    const unsigned long long mask = okA ? maskA : maskB;

    // EXEC
    double *dE;
    CHECK(hipMalloc(&dE, 64 * 8));
    std::vector<double> hE(64);
    for (int inf_lane : {-1, 48, 40}) {
        probe_exec<<<1, 64>>>(dE, maskA, inf_lane);
        CHECK(hipMemcpy(hE.data(), dE, 64 * 8, hipMemcpyDeviceToHost));
        int written_hi = 0, nan_lo = 0, good_lo = 0;
        for (int l = 0; l < 64; ++l) {
            if (l >= 32 && hE[l] != -777.0) ++written_hi;
            if (l < 32) { if (std::isnan(hE[l])) ++nan_lo; else if (hE[l] == l + 1 + 0.5) ++good_lo; }
        }
        printf("   EXEC = lanes 0..31, inf in B of lane %d: lanes >= 32 written %d/32; lanes < 32 correct %d, NaN %d  (D[0]=%g D[8]=%g D[40]=%g)\n",
               inf_lane, written_hi, good_lo, nan_lo, hE[0], hE[8], hE[40]);
    }

    // ---- B
    const int n = 1 << 22;
    std::vector<double> hx(n), hc(n), r1(n), r2(n);
    double *dx, *dc, *d1, *d2;
    CHECK(hipMalloc(&dx, n * 8)); CHECK(hipMalloc(&dc, n * 8)); CHECK(hipMalloc(&d1, n * 8)); CHECK(hipMalloc(&d2, n * 8));
    printf("B. D = pattern(s) x X + C against v_fma_f64(s, X, C), %d operands per row, bitwise:\n", n);
    for (int cls = 0; cls < 5; ++cls) {
        for (double s : {1.0, -1.0, 2.0}) {
            for (int i = 0; i < n; ++i) {
                if (cls < 3) { hx[i] = gen(cls); hc[i] = gen(cls == 2 ? (int)(rnd() % 3) : cls); }
                else if (cls == 3) { hx[i] = gen(3); hc[i] = -s * hx[i] * (1.0 + (double)((int)(rnd() % 5) - 2) * 0x1p-52); }
                else { hx[i] = gen(0); hc[i] = gen(2); }
            }
            CHECK(hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(dc, hc.data(), n * 8, hipMemcpyHostToDevice));
            exact_kernel<<<1024, 256>>>(dx, dc, d1, d2, mask, s, data_in_a, n);
            CHECK(hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost));
            long bad = 0, bad_host = 0, sub_out = 0, first = -1;
            for (int i = 0; i < n; ++i) {
                if (to_bits(r1[i]) != to_bits(r2[i])) { if (first < 0) first = i; ++bad; }
                if (to_bits(r2[i]) != to_bits(fma(s, hx[i], hc[i]))) ++bad_host;
                if (r2[i] != 0.0 && std::fabs(r2[i]) < 2.2250738585072014e-308) ++sub_out;
            }
            static const char *cn[5] = {"orbit-like |x|<4", "any finite exponent", "subnormal X", "cancelling X, C", "normal X, subnormal C"};
            printf("   %-22s s=%+.0f: MFMA != VALU in %ld, VALU != host fma in %ld (subnormal results: %ld)", cn[cls], s, bad, bad_host, sub_out);
            if (first >= 0) printf("   first: x=%a c=%a mfma=%a valu=%a", hx[first], hc[first], r1[first], r2[first]);
            printf("\n");
        }
    }

    // ---- C
    const int trips = 4000;   // x 16 steps
    double *d_out;
    unsigned long long *d_ticks;
    printf("C. loop body, %d trips x 16 steps per wave; ns and shader ticks per wave-step per SIMD (lower = better), results vs V0:\n", trips);
    for (int wpc : {8, 4}) {   // workgroups of 4 waves per CU -> waves per SIMD
        const int wgs = cus * wpc;
        CHECK(hipMalloc(&d_out, (size_t)wgs * 256 * 16));
        CHECK(hipMalloc(&d_ticks, wgs * 8));
        std::vector<double> ref((size_t)wgs * 512), got((size_t)wgs * 512);
        printf("   -- %d waves per SIMD (%d workgroups of 256)\n", wpc, wgs);
        double base_ns = 0;
        for (int v = 0; v < NVAR; ++v) {
            double tick = 0, ms = 0;
            switch (v) {
            case V0: ms = run_variant<V0>(wgs, trips, d_out, d_ticks, mask, &tick); break;
            case V1_N4: ms = run_variant<V1_N4>(wgs, trips, d_out, d_ticks, mask, &tick); break;
            case V1_N2: ms = run_variant<V1_N2>(wgs, trips, d_out, d_ticks, mask, &tick); break;
            case V1_N0: ms = run_variant<V1_N0>(wgs, trips, d_out, d_ticks, mask, &tick); break;
            case V2_N4: ms = run_variant<V2_N4>(wgs, trips, d_out, d_ticks, mask, &tick); break;
            case V3_N5: ms = run_variant<V3_N5>(wgs, trips, d_out, d_ticks, mask, &tick); break;
            case V4_N4: ms = run_variant<V4_N4>(wgs, trips, d_out, d_ticks, mask, &tick); break;
            case V5: ms = run_variant<V5>(wgs, trips, d_out, d_ticks, mask, &tick); break;
            case V6_N5: ms = run_variant<V6_N5>(wgs, trips, d_out, d_ticks, mask, &tick); break;
            case V7_N4: ms = run_variant<V7_N4>(wgs, trips, d_out, d_ticks, mask, &tick); break;
            }
            CHECK(hipMemcpy(got.data(), d_out, got.size() * 8, hipMemcpyDeviceToHost));
            if (v == V0) ref = got;
            long diff = 0;
            for (size_t i = 0; i < got.size(); ++i) if (to_bits(got[i]) != to_bits(ref[i])) ++diff;
            const double wave_steps = (double)wgs * 4.0 * trips * 16.0;
            const double ns = ms * 1e6 * (cus * 4.0) / wave_steps;   // per wave-step per SIMD
            if (v == V0) base_ns = ns;
            const bool comparable = v == V0 || v == V1_N4 || v == V1_N0 || v == V1_N2 || v == V2_N4 || v == V3_N5 || v == V4_N4;
            printf("   %-42s %8.3f ms  %6.2f ns/wave-step/SIMD (x%.3f of V0)  wave ticks/step %7.2f  %s\n", kNames[v], ms, ns, ns / base_ns,
                   tick / (trips * 16.0), comparable ? (diff ? "RESULTS DIFFER from V0" : "bit-identical to V0") : "");
            if (comparable && diff) printf("      (%ld of %zu values differ)\n", diff, got.size());
        }
        CHECK(hipFree(d_out));
        CHECK(hipFree(d_ticks));
    }
    return 0;
}

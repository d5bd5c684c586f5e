// valu_issue.hip -- what bounds the fp64 issue rate of the escape loop on gfx950?  
//
// The product's strict loop spends 6 fp64 VALU instructions per step (v_add, v_mul, v_add, v_fma, v_mul, v_mul) and
// its counters say the vector ALU is busy 92 % of the cycles with 8 waves per SIMD: 4.3-4.4 shader cycles per
// wave-instruction where the 16-lane fp64 pipe needs 4.0.  This file measures that body in SHADER CYCLES (s_memtime;
// the governor is out of the picture) while varying one thing at a time:
//   * the physical VGPRs of the operands (the register file's four banks: every pair of sources in different banks,
//     the compiler-like packing, every source in the same bank);
//   * the distance between dependent instructions (1, 2 or 4 pixels per lane, interleaved);
//   * waves per SIMD (4, 6, 8), with and without staggered s_setprio;
//   * s_nop between the VALU instructions; the product's group structure (add + compare + branch per 16 steps);
//   * 4-byte against 8-byte encodings (fp32 v_mul_f32 e32 / e64: is instruction fetch part of it?).
// Every kernel is one asm block on hard-coded registers; single-wave workgroups like the product; the grid is exactly
// CUs x 4 SIMDs x W waves, and each wave records where it ran (HW_ID / XCC_ID), so the number of waves that shared its
// SIMD is measured, not assumed.  Cycles per SIMD-instruction = slope of a wave's s_memtime span between two trip
// counts / (its instructions per trip x the waves on its SIMD).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue profiles/microbench/valu_issue.hip && /tmp/valu_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <unistd.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Rec { unsigned long long t0, t1, r0, r1; unsigned hw_id, xcc_id; };   // s_memtime / s_memrealtime (100 MHz) at start and end

// one step on the register set (ZR, ZI, A, B) with temporaries (T, P) and the constants (CR, CI)
#define STEP(ZR, ZI, A, B, T, P, CR, CI)                  \
    "v_add_f64 " T ", " A ", -" B "\n"                     \
    "v_mul_f64 " P ", " ZR ", " ZI "\n"                    \
    "v_add_f64 " ZR ", " T ", " CR "\n"                    \
    "v_fma_f64 " ZI ", " P ", 2.0, " CI "\n"               \
    "v_mul_f64 " A ", " ZR ", " ZR "\n"                    \
    "v_mul_f64 " B ", " ZI ", " ZI "\n"
#define STEP_NOP(ZR, ZI, A, B, T, P, CR, CI)              \
    "v_add_f64 " T ", " A ", -" B "\ns_nop 0\n"            \
    "v_mul_f64 " P ", " ZR ", " ZI "\ns_nop 0\n"           \
    "v_add_f64 " ZR ", " T ", " CR "\ns_nop 0\n"           \
    "v_fma_f64 " ZI ", " P ", 2.0, " CI "\ns_nop 0\n"      \
    "v_mul_f64 " A ", " ZR ", " ZR "\ns_nop 0\n"           \
    "v_mul_f64 " B ", " ZI ", " ZI "\ns_nop 0\n"
// two / four pixels interleaved instruction by instruction (dependent distance 2 / 4)
#define STEP2(R0, R1)                                                                                   \
    "v_add_f64 " R0(T) ", " R0(A) ", -" R0(B) "\n"   "v_add_f64 " R1(T) ", " R1(A) ", -" R1(B) "\n"       \
    "v_mul_f64 " R0(P) ", " R0(ZR) ", " R0(ZI) "\n"  "v_mul_f64 " R1(P) ", " R1(ZR) ", " R1(ZI) "\n"      \
    "v_add_f64 " R0(ZR) ", " R0(T) ", " R0(CR) "\n"  "v_add_f64 " R1(ZR) ", " R1(T) ", " R1(CR) "\n"      \
    "v_fma_f64 " R0(ZI) ", " R0(P) ", 2.0, " R0(CI) "\n" "v_fma_f64 " R1(ZI) ", " R1(P) ", 2.0, " R1(CI) "\n" \
    "v_mul_f64 " R0(A) ", " R0(ZR) ", " R0(ZR) "\n"  "v_mul_f64 " R1(A) ", " R1(ZR) ", " R1(ZR) "\n"      \
    "v_mul_f64 " R0(B) ", " R0(ZI) ", " R0(ZI) "\n"  "v_mul_f64 " R1(B) ", " R1(ZI) ", " R1(ZI) "\n"
#define STEP4(R0, R1, R2, R3)                                                                           \
    "v_add_f64 " R0(T) ", " R0(A) ", -" R0(B) "\n"   "v_add_f64 " R1(T) ", " R1(A) ", -" R1(B) "\n"       \
    "v_add_f64 " R2(T) ", " R2(A) ", -" R2(B) "\n"   "v_add_f64 " R3(T) ", " R3(A) ", -" R3(B) "\n"       \
    "v_mul_f64 " R0(P) ", " R0(ZR) ", " R0(ZI) "\n"  "v_mul_f64 " R1(P) ", " R1(ZR) ", " R1(ZI) "\n"      \
    "v_mul_f64 " R2(P) ", " R2(ZR) ", " R2(ZI) "\n"  "v_mul_f64 " R3(P) ", " R3(ZR) ", " R3(ZI) "\n"      \
    "v_add_f64 " R0(ZR) ", " R0(T) ", " R0(CR) "\n"  "v_add_f64 " R1(ZR) ", " R1(T) ", " R1(CR) "\n"      \
    "v_add_f64 " R2(ZR) ", " R2(T) ", " R2(CR) "\n"  "v_add_f64 " R3(ZR) ", " R3(T) ", " R3(CR) "\n"      \
    "v_fma_f64 " R0(ZI) ", " R0(P) ", 2.0, " R0(CI) "\n" "v_fma_f64 " R1(ZI) ", " R1(P) ", 2.0, " R1(CI) "\n" \
    "v_fma_f64 " R2(ZI) ", " R2(P) ", 2.0, " R2(CI) "\n" "v_fma_f64 " R3(ZI) ", " R3(P) ", 2.0, " R3(CI) "\n" \
    "v_mul_f64 " R0(A) ", " R0(ZR) ", " R0(ZR) "\n"  "v_mul_f64 " R1(A) ", " R1(ZR) ", " R1(ZR) "\n"      \
    "v_mul_f64 " R2(A) ", " R2(ZR) ", " R2(ZR) "\n"  "v_mul_f64 " R3(A) ", " R3(ZR) ", " R3(ZR) "\n"      \
    "v_mul_f64 " R0(B) ", " R0(ZI) ", " R0(ZI) "\n"  "v_mul_f64 " R1(B) ", " R1(ZI) ", " R1(ZI) "\n"      \
    "v_mul_f64 " R2(B) ", " R2(ZI) ", " R2(ZI) "\n"  "v_mul_f64 " R3(B) ", " R3(ZI) ", " R3(ZI) "\n"

// register maps: name -> physical pair.  Bank of a 64-bit operand = (low register) mod 4.
// PACKED: consecutive pairs (what a compiler does).  zr 0, zi 2, a 0, b 2, t 0, p 2, cr 0, ci 2: zr = t + cr reads banks (0, 0),
//         zi = fma(p, 2, ci) reads (2, 2).
#define PK_ZR "v[0:1]"
#define PK_ZI "v[2:3]"
#define PK_A "v[4:5]"
#define PK_B "v[6:7]"
#define PK_T "v[8:9]"
#define PK_P "v[10:11]"
#define PK_CR "v[12:13]"
#define PK_CI "v[14:15]"
#define PK(x) PK_##x
// SPREAD: the two register sources of every instruction in different banks (t 0 / cr 2, p 0 / ci 2)
#define SP_ZR "v[0:1]"
#define SP_ZI "v[2:3]"
#define SP_A "v[4:5]"
#define SP_B "v[6:7]"
#define SP_T "v[8:9]"
#define SP_P "v[12:13]"
#define SP_CR "v[10:11]"
#define SP_CI "v[14:15]"
#define SP(x) SP_##x
// SAME: every operand in bank 0
#define SM_ZR "v[0:1]"
#define SM_ZI "v[4:5]"
#define SM_A "v[8:9]"
#define SM_B "v[12:13]"
#define SM_T "v[16:17]"
#define SM_P "v[20:21]"
#define SM_CR "v[24:25]"
#define SM_CI "v[28:29]"
#define SM(x) SM_##x
// (pairs must start at an even register on gfx90a+: "vgpr tuples must be 64 bit aligned" -- a 64-bit operand lives in banks 0+1 or 2+3)
// the interleaved bodies (2 / 4 pixels per lane): pixel k in v[16k .. 16k+11]; the pixels share c (timing only), cr in bank 2 and
// ci in bank 0, so that the two register sources of every instruction sit in different banks, as in SPREAD
#define M0_ZR "v[0:1]"
#define M0_ZI "v[2:3]"
#define M0_A "v[4:5]"
#define M0_B "v[6:7]"
#define M0_T "v[8:9]"
#define M0_P "v[14:15]"
#define M0_CR "v[10:11]"
#define M0_CI "v[12:13]"
#define M0(x) M0_##x
#define Q1_ZR "v[16:17]"
#define Q1_ZI "v[18:19]"
#define Q1_A "v[20:21]"
#define Q1_B "v[22:23]"
#define Q1_T "v[24:25]"
#define Q1_P "v[26:27]"
#define Q1_CR "v[10:11]"
#define Q1_CI "v[12:13]"
#define Q1(x) Q1_##x
#define Q2_ZR "v[28:29]"
#define Q2_ZI "v[30:31]"
#define Q2_A "v[32:33]"
#define Q2_B "v[34:35]"
#define Q2_T "v[36:37]"
#define Q2_P "v[38:39]"
#define Q2_CR "v[10:11]"
#define Q2_CI "v[12:13]"
#define Q2(x) Q2_##x
#define Q3_ZR "v[40:41]"
#define Q3_ZI "v[42:43]"
#define Q3_A "v[44:45]"
#define Q3_B "v[46:47]"
#define Q3_T "v[48:49]"
#define Q3_P "v[50:51]"
#define Q3_CR "v[10:11]"
#define Q3_CI "v[12:13]"
#define Q3(x) Q3_##x

#define S1(M) STEP(M(ZR), M(ZI), M(A), M(B), M(T), M(P), M(CR), M(CI))
#define S1N(M) STEP_NOP(M(ZR), M(ZI), M(A), M(B), M(T), M(P), M(CR), M(CI))
#define X4(s) s s s s
#define X16(s) X4(X4(s))

// initialise pixel M: c = (cr, ci) from SGPR pairs, z = c, a = zr^2, b = zi^2
#define LOAD_C(M, CRS, CIS)                                                                 \
    "v_lshrrev_b64 " M(CR) ", 0, " CRS "\n" "v_lshrrev_b64 " M(CI) ", 0, " CIS "\n"          \
    "v_lshrrev_b64 " M(ZR) ", 0, " CRS "\n" "v_lshrrev_b64 " M(ZI) ", 0, " CIS "\n"          \
    "v_mul_f64 " M(A) ", " M(ZR) ", " M(ZR) "\n" "v_mul_f64 " M(B) ", " M(ZI) ", " M(ZI) "\n"

#define CLOB32 "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"
#define CLOB52 "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51"

// the timed loop: PRE (register setup), then t0, `trips` x BODY, t1; SINK = the pair whose value is written out
#define TIMED(PRE, BODY)                                   \
    PRE                                                    \
    "s_waitcnt lgkmcnt(0)\n"                               \
    "s_memrealtime %[r0]\n"                                \
    "s_memtime %[t0]\n"                                    \
    "s_waitcnt lgkmcnt(0)\n"                               \
    ".Lloop_%=:\n"                                         \
    BODY                                                   \
    "s_sub_u32 %[n], %[n], 1\n"                            \
    "s_cmp_lg_u32 %[n], 0\n"                               \
    "s_cbranch_scc1 .Lloop_%=\n"                           \
    "s_memtime %[t1]\n"                                    \
    "s_memrealtime %[r1]\n"                                \
    "s_waitcnt lgkmcnt(0)\n"

enum Variant { V_PACKED, V_SPREAD, V_SAME, V_NOP, V_PROD, V_2PX, V_4PX, V_F32_E32, V_F32_E64, V_COUNT };
static const char *kNames[V_COUNT] = {"packed", "spread", "same-bank", "packed+s_nop", "product-group16", "2 px/lane", "4 px/lane",
                                      "v_mul_f32 e32 (4 B)", "v_mul_f32 e64 (8 B)"};
// VALU instructions per loop trip
static const int kInstr[V_COUNT] = {96, 96, 96, 96, 2 * (96 + 2), 192, 384, 96, 96};

template <int V>
__global__ __launch_bounds__(64) void issue_kernel(Rec *out, double *sink, double cr_in, double ci_in, unsigned trips, unsigned prio)
{
    unsigned long long t0, t1, r0, r1;
    unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)trips);
    const unsigned long long crs = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(cr_in) & 0xffffffffll)) |
                                   ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(cr_in) >> 32)) << 32);
    const unsigned long long cis = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(ci_in) & 0xffffffffll)) |
                                   ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(__double_as_longlong(ci_in) >> 32)) << 32);
    if (prio == 1) {   // staggered priorities: wave slot parity decides (s_setprio takes an immediate)
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        if (hw & 1u) asm volatile("s_setprio 2"); else asm volatile("s_setprio 0");
    } else if (prio == 2) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        switch (hw & 3u) { case 0: asm volatile("s_setprio 0"); break; case 1: asm volatile("s_setprio 1"); break;
                           case 2: asm volatile("s_setprio 2"); break; default: asm volatile("s_setprio 3"); break; }
    }
    double res;
#define OPERANDS_(CLOB) : [t0] "=&s"(t0), [t1] "=&s"(t1), [r0] "=&s"(r0), [r1] "=&s"(r1), [n] "+s"(n), [res] "=v"(res) : [crs] "s"(crs), [cis] "s"(cis) : "vcc", "scc", "memory", CLOB
#define OPERANDS OPERANDS_(CLOB32)
#define OPERANDS_MULTI OPERANDS_(CLOB52)
    if (V == V_PACKED) {
        asm volatile(TIMED(LOAD_C(PK, "%[crs]", "%[cis]"), X16(S1(PK))) "v_lshrrev_b64 %[res], 0, " PK_ZR "\n" OPERANDS);
    } else if (V == V_SPREAD) {
        asm volatile(TIMED(LOAD_C(SP, "%[crs]", "%[cis]"), X16(S1(SP))) "v_lshrrev_b64 %[res], 0, " SP_ZR "\n" OPERANDS);
    } else if (V == V_SAME) {
        asm volatile(TIMED(LOAD_C(SM, "%[crs]", "%[cis]"), X16(S1(SM))) "v_lshrrev_b64 %[res], 0, " SM_ZR "\n" OPERANDS);
    } else if (V == V_NOP) {
        asm volatile(TIMED(LOAD_C(PK, "%[crs]", "%[cis]"), X16(S1N(PK))) "v_lshrrev_b64 %[res], 0, " PK_ZR "\n" OPERANDS);
    } else if (V == V_PROD) {
        // the product's 16-step group: 16 steps, then m = a + b, the NaN-inclusive compare and a branch that is never taken
        asm volatile(TIMED(LOAD_C(PK, "%[crs]", "%[cis]"),
                           X16(S1(PK)) "v_add_f64 v[16:17], " PK_A ", " PK_B "\nv_cmp_ngt_f64 vcc, 4.0, v[16:17]\ns_cbranch_vccnz .Lout_%=\n"
                           X16(S1(PK)) "v_add_f64 v[16:17], " PK_A ", " PK_B "\nv_cmp_ngt_f64 vcc, 4.0, v[16:17]\ns_cbranch_vccnz .Lout_%=\n")
                     ".Lout_%=:\n" "v_lshrrev_b64 %[res], 0, " PK_ZR "\n" OPERANDS);
    } else if (V == V_2PX) {
        asm volatile(TIMED(LOAD_C(M0, "%[crs]", "%[cis]") LOAD_C(Q1, "%[crs]", "%[cis]"), X16(STEP2(M0, Q1)))
                     "v_add_f64 %[res], " M0_ZR ", " Q1_ZR "\n" OPERANDS_MULTI);
    } else if (V == V_4PX) {
        asm volatile(TIMED(LOAD_C(M0, "%[crs]", "%[cis]") LOAD_C(Q1, "%[crs]", "%[cis]") LOAD_C(Q2, "%[crs]", "%[cis]") LOAD_C(Q3, "%[crs]", "%[cis]"),
                           X16(STEP4(M0, Q1, Q2, Q3)))
                     "v_add_f64 %[res], " M0_ZR ", " Q1_ZR "\nv_add_f64 %[res], %[res], " Q2_ZR "\nv_add_f64 %[res], %[res], " Q3_ZR "\n" OPERANDS_MULTI);
    } else if (V == V_F32_E32) {
#define M6 "v_mul_f32_e32 v0, v8, v0\nv_mul_f32_e32 v1, v9, v1\nv_mul_f32_e32 v2, v10, v2\nv_mul_f32_e32 v3, v11, v3\nv_mul_f32_e32 v4, v8, v4\nv_mul_f32_e32 v5, v9, v5\n"
        asm volatile(TIMED("v_mov_b32 v0, 1.0\nv_mov_b32 v1, 1.0\nv_mov_b32 v2, 1.0\nv_mov_b32 v3, 1.0\nv_mov_b32 v4, 1.0\nv_mov_b32 v5, 1.0\n"
                           "v_mov_b32 v8, 1.0\nv_mov_b32 v9, 1.0\nv_mov_b32 v10, 1.0\nv_mov_b32 v11, 1.0\n", X16(M6))
                     "v_cvt_f64_f32 %[res], v0\n" OPERANDS);
#undef M6
    } else {
#define M6 "v_mul_f32_e64 v0, v8, v0\nv_mul_f32_e64 v1, v9, v1\nv_mul_f32_e64 v2, v10, v2\nv_mul_f32_e64 v3, v11, v3\nv_mul_f32_e64 v4, v8, v4\nv_mul_f32_e64 v5, v9, v5\n"
        asm volatile(TIMED("v_mov_b32 v0, 1.0\nv_mov_b32 v1, 1.0\nv_mov_b32 v2, 1.0\nv_mov_b32 v3, 1.0\nv_mov_b32 v4, 1.0\nv_mov_b32 v5, 1.0\n"
                           "v_mov_b32 v8, 1.0\nv_mov_b32 v9, 1.0\nv_mov_b32 v10, 1.0\nv_mov_b32 v11, 1.0\n", X16(M6))
                     "v_cvt_f64_f32 %[res], v0\n" OPERANDS);
#undef M6
    }
    unsigned hw_id, xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    if (threadIdx.x == 0) {
        out[blockIdx.x].t0 = t0;
        out[blockIdx.x].t1 = t1;
        out[blockIdx.x].r0 = r0;
        out[blockIdx.x].r1 = r1;
        out[blockIdx.x].hw_id = hw_id;
        out[blockIdx.x].xcc_id = xcc_id;
    }
    sink[(size_t)blockIdx.x * 64 + threadIdx.x] = res;
}

struct Result { double cyc_per_simd_instr, cyc_per_wave_instr, span_cyc_per_simd_instr, ns_per_simd_instr, waves_med, tick_mhz; int waves_min, waves_max; };

static bool g_dump = false;   // print the (start, end) of the waves of one SIMD: do they run side by side or one pair after the other?

template <int V>
static Result run(int cus, int waves_per_simd, unsigned prio)
{
    const int nwaves = cus * 4 * waves_per_simd;
    Rec *d_rec; double *d_sink;
    CHECK(hipMalloc(&d_rec, nwaves * sizeof(Rec)));
    CHECK(hipMalloc(&d_sink, (size_t)nwaves * 64 * sizeof(double)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const unsigned trips[2] = {V == V_4PX ? 100u : V == V_2PX ? 200u : 400u, V == V_4PX ? 300u : V == V_2PX ? 600u : 1200u};
    std::vector<Rec> h[2];
    float ms[2];
    for (int rep = 0; rep < 2; ++rep) {
        for (int pass = 0; pass < 2; ++pass) {   // first pass warms the clock and the instruction cache
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(issue_kernel<V>, dim3(nwaves), dim3(64), 0, 0, d_rec, d_sink, -0.1, 0.2, trips[rep], prio);
            CHECK(hipEventRecord(e1));
            CHECK(hipDeviceSynchronize());
        }
        CHECK(hipEventElapsedTime(&ms[rep], e0, e1));
        h[rep].resize(nwaves);
        CHECK(hipMemcpy(h[rep].data(), d_rec, nwaves * sizeof(Rec), hipMemcpyDeviceToHost));
    }
    // waves per SIMD from the placement census (key: XCC, and HW_ID without wave slot / pipe / queue bits)
    auto key = [](const Rec &r) { return ((unsigned long long)(r.xcc_id & 0xfu) << 32) | (r.hw_id & 0x0000ff30u); };
    std::map<unsigned long long, int> per_simd;
    std::map<unsigned long long, std::pair<unsigned long long, unsigned long long>> span[2];   // first start, last end per SIMD
    for (int rep = 0; rep < 2; ++rep)
        for (const Rec &r : h[rep]) {
            if (rep == 1) per_simd[key(r)]++;
            auto it = span[rep].find(key(r));
            if (it == span[rep].end()) span[rep][key(r)] = {r.t0, r.t1};
            else { it->second.first = std::min(it->second.first, r.t0); it->second.second = std::max(it->second.second, r.t1); }
        }
    std::vector<double> per_wave, per_simd_instr, wcount, span_instr, mhz;
    const double d_instr = (double)(trips[1] - trips[0]) * kInstr[V];
    for (int w = 0; w < nwaves; ++w) {
        const double slope = ((double)(h[1][w].t1 - h[1][w].t0) - (double)(h[0][w].t1 - h[0][w].t0)) / d_instr;
        const int share = per_simd[key(h[1][w])];
        per_wave.push_back(slope);
        per_simd_instr.push_back(slope / share);
        wcount.push_back(share);
        if (h[1][w].r1 > h[1][w].r0) mhz.push_back(100.0 * (double)(h[1][w].t1 - h[1][w].t0) / (double)(h[1][w].r1 - h[1][w].r0));
    }
    // per SIMD: (last end - first start) of the long run minus the same of the short run, over the instructions all its
    // waves issued in between: shader cycles per SIMD-instruction whatever order the arbiter ran the waves in
    for (auto &kv : span[1]) {
        auto it = span[0].find(kv.first);
        if (it == span[0].end() || per_simd[kv.first] == 0) continue;
        const double d = (double)(kv.second.second - kv.second.first) - (double)(it->second.second - it->second.first);
        span_instr.push_back(d / (d_instr * per_simd[kv.first]));
    }
    auto med = [](std::vector<double> v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    Result r;
    r.cyc_per_wave_instr = med(per_wave);
    r.cyc_per_simd_instr = med(per_simd_instr);
    r.span_cyc_per_simd_instr = med(span_instr);
    r.waves_med = med(wcount);
    r.waves_min = (int)*std::min_element(wcount.begin(), wcount.end());
    r.waves_max = (int)*std::max_element(wcount.begin(), wcount.end());
    r.ns_per_simd_instr = (ms[1] - ms[0]) * 1e6 / (d_instr * waves_per_simd);
    r.tick_mhz = med(mhz);
    if (g_dump) {
        const unsigned long long k0 = key(h[1][nwaves / 2]);
        unsigned long long base = ~0ull;
        for (const Rec &x : h[1]) if (key(x) == k0) base = std::min(base, x.t0);
        printf("    the waves of one SIMD (long run), s_memtime ticks from the first start:");
        std::vector<std::pair<unsigned long long, unsigned long long>> iv;
        for (const Rec &x : h[1]) if (key(x) == k0) iv.push_back({x.t0 - base, x.t1 - base});
        std::sort(iv.begin(), iv.end());
        for (auto &p : iv) printf("  [%llu .. %llu]", p.first, p.second);
        printf("\n");
    }
    CHECK(hipFree(d_rec)); CHECK(hipFree(d_sink));
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return r;
}

template <int V>
static void report(int cus)
{
    for (int w : {8, 6, 4, 2, 1}) {
        for (unsigned prio : {0u, 1u, 2u}) {
            if (prio != 0 && (w != 8 || (V != V_PACKED && V != V_PROD))) continue;
            if (w < 4 && V != V_PACKED && V != V_PROD && V != V_4PX) continue;
            g_dump = (V == V_PACKED || V == V_PROD) && prio == 0 && (w == 8 || w == 4);
            const Result r = run<V>(cus, w, prio);
            printf("%-22s waves/SIMD %d (placed: median %.0f, min %d, max %d) setprio %-18s | per wave %.2f ticks/instr | per SIMD (first start .. last end) "
                   "%.3f ticks/instr | s_memtime ticks at %.0f MHz (vs s_memrealtime) | kernel-time slope %.3f ns per SIMD-instr\n",
                   kNames[V], w, r.waves_med, r.waves_min, r.waves_max, prio == 0 ? "none" : prio == 1 ? "0/2 by slot parity" : "0..3 by slot",
                   r.cyc_per_wave_instr, r.span_cyc_per_simd_instr, r.tick_mhz, r.ns_per_simd_instr);
            fflush(stdout);
        }
    }
}

// ---- clock probe: what does the shader clock do while ANOTHER process (bench.py) loads the chip? -------------------
// One wave reads s_memtime (shader cycles) and s_memrealtime (100 MHz) ~100 us apart; the ratio is the shader clock
// the wave lived through.  Run beside a benchmark:   /tmp/valu_issue --probe 8000 &  python bench.py ...
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long *out, unsigned spins)
{
    unsigned long long t0, t1, r0, r1;
    unsigned n = spins;
    asm volatile("s_memrealtime %[r0]\n"
                 "s_memtime %[t0]\n"
                 "s_waitcnt lgkmcnt(0)\n"
                 ".Lspin_%=:\n"
                 "s_sleep 127\n"
                 "s_sub_u32 %[n], %[n], 1\n"
                 "s_cmp_lg_u32 %[n], 0\n"
                 "s_cbranch_scc1 .Lspin_%=\n"
                 "s_memtime %[t1]\n"
                 "s_memrealtime %[r1]\n"
                 "s_waitcnt lgkmcnt(0)\n"
                 : [t0] "=&s"(t0), [t1] "=&s"(t1), [r0] "=&s"(r0), [r1] "=&s"(r1), [n] "+s"(n) : : "scc", "memory");
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

static int probe(int total_ms)
{
    unsigned long long *d, h[2];
    CHECK(hipMalloc(&d, 16));
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<double> mhz;
    printf("# ms since start, shader clock MHz (s_memtime / s_memrealtime over the probe's life), probe kernel ms (launch to end: long = the chip was full)\n");
    hipEvent_t start; CHECK(hipEventCreate(&start)); CHECK(hipEventRecord(start, s));
    for (;;) {
        CHECK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s, d, 30u);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s));
        CHECK(hipStreamSynchronize(s));
        float since = 0, dur = 0;
        CHECK(hipEventElapsedTime(&since, start, e1));
        CHECK(hipEventElapsedTime(&dur, e0, e1));
        const double m = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
        printf("%8.1f %7.1f %6.3f\n", since, m, dur);
        mhz.push_back(m);
        if (since > total_ms) break;
        usleep(8000);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 2 && std::string(argv[1]) == "--probe") return probe(atoi(argv[2]));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s arch %s CUs %d clock %d MHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    const int cus = prop.multiProcessorCount;
    report<V_PACKED>(cus);
    report<V_SPREAD>(cus);
    report<V_SAME>(cus);
    report<V_NOP>(cus);
    report<V_PROD>(cus);
    report<V_2PX>(cus);
    report<V_4PX>(cus);
    report<V_F32_E32>(cus);
    report<V_F32_E64>(cus);
    return 0;
}

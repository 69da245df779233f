// Micro-benchmark: VALU issue cost by instruction FORM on MI355X (gfx950), at 1..8 waves per SIMD.
// Each kernel runs ITER iterations of 16 instructions of one form over 8 independent registers; the
// figure printed is SIMD cycles per wave-instruction at the nominal 2.4 GHz (2 = the documented fp32
// rate with >= 2 waves per SIMD, 4 = half rate).
//   hipcc --offload-arch=gfx950 -O3 valu_issue_bench.hip -o valu_issue_bench && ./valu_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITER 4096
// I(d, a) expands to one instruction string using destination/source register index d and partner a
#define BODY8(I) I("%0", "%1") I("%1", "%2") I("%2", "%3") I("%3", "%4") I("%4", "%5") I("%5", "%6") I("%6", "%7") I("%7", "%0")
#define BODY16(I) BODY8(I) BODY8(I)
#define OPERANDS : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c), "v"(d), "s"(sa), "s"(sb) : "vcc", "scc", "s20", "s21"

#define F_FMA_VVV(D, A)     "v_fma_f32 " D ", " D ", %8, %9\n"
#define F_FMA_SQ(D, A)      "v_fma_f32 " D ", %8, %8, " D "\n"            /* x*x + acc : two distinct VGPRs */
#define F_FMA_VSV(D, A)     "v_fma_f32 " D ", " D ", %10, %9\n"
#define F_FMA_LIT(D, A)     "v_fma_f32 " D ", " D ", 2.0, %9\n"            /* inline constant */
#define F_FMAC_VV(D, A)     "v_fmac_f32 " D ", %8, %9\n"                   /* VOP2: D += a*b */
#define F_FMAC_SV(D, A)     "v_fmac_f32 " D ", %10, %9\n"
#define F_FMAAK(D, A)       "v_fmaak_f32 " D ", " D ", %8, 0x3f7fbe77\n"   /* VOP2 with literal addend */
#define F_MUL_VV(D, A)      "v_mul_f32 " D ", %8, " D "\n"
#define F_MUL_SV(D, A)      "v_mul_f32 " D ", %10, " D "\n"
#define F_MUL_LIT(D, A)     "v_mul_f32 " D ", 0x3f7fbe77, " D "\n"
#define F_ADD_VV(D, A)      "v_add_f32 " D ", " A ", " D "\n"
#define F_SUB_ABS_V(D, A)   "v_sub_f32 " D ", |" D "|, %8\n"              /* VOP3 encoding, 2 VGPRs */
#define F_SUB_ABS_S(D, A)   "v_sub_f32 " D ", |" D "|, %10\n"
#define F_MAX_VV(D, A)      "v_max_f32 " D ", %8, " D "\n"
#define F_MAX3(D, A)        "v_max3_f32 " D ", " D ", %8, %9\n"
#define F_MAXIMUM3(D, A)    "v_maximum3_f32 " D ", " D ", %8, %9\n"
#define F_MAXIMUM3_2(D, A)  "v_maximum3_f32 " D ", " D ", %8, %8\n"       /* two distinct VGPRs */
#define F_CMP_VCC(D, A)     "v_cmp_gt_f32 vcc, " D ", %8\n"
#define F_CMP_S(D, A)       "v_cmp_gt_f32 s[20:21], " D ", %8\n"
#define F_CND_VCC(D, A)     "v_cndmask_b32 " D ", " D ", " A ", vcc\n"
#define F_CMPCND(D, A)      "v_cmp_gt_f32 vcc, " D ", %8\n v_cndmask_b32 " D ", " D ", " A ", vcc\n"   /* 2 instr */
#define F_MOV(D, A)         "v_mov_b32 " D ", " A "\n"
#define F_SQRT(D, A)        "v_sqrt_f32 " D ", " D "\n"
#define F_RCP(D, A)         "v_rcp_f32 " D ", " D "\n"
#define F_PK_MUL(D, A)      "v_pk_mul_f32 %[p0], %[p0], %[p1]\n"
#define F_PK_FMA(D, A)      "v_pk_fma_f32 %[p0], %[p0], %[p1], %[p2]\n"
#define F_CHAIN(D, A)       "v_fma_f32 %0, %0, %8, %9\n"
#define F_MUL_CHAIN(D, A)   "v_mul_f32 %0, %8, %0\n"
#define F_VALU_SALU(D, A)   "v_mul_f32 " D ", %8, " D "\n s_add_u32 s20, s20, 1\n"
#define F_VALU_SALU_S(D, A) "v_mul_f32 " D ", %10, " D "\n s_add_u32 s20, s20, 1\n"
#define F_AND(D, A)         "v_and_b32 " D ", 0x7fffffff, " D "\n"
#define F_ADD_I(D, A)       "v_add_u32 " D ", 1, " D "\n"
#define F_BFE(D, A)         "v_add_u32 " D ", %10, " D "\n"

template <int FORM>
__global__ void __launch_bounds__(256) k(float* out, float sa, float sb) {
  float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  float c = 0.999f + 1e-6f * threadIdx.x, d = 1e-3f;
  typedef float float2v __attribute__((ext_vector_type(2)));
  float2v p0 = {v0, v1}, p1 = {c, c}, p2 = {d, d};
  for (int i = 0; i < ITER; ++i) {
#define CASE(N, I) if constexpr (FORM == N) asm volatile(BODY16(I) OPERANDS);
    CASE(0, F_FMA_VVV) CASE(1, F_FMA_SQ) CASE(2, F_FMA_VSV) CASE(3, F_FMA_LIT) CASE(4, F_FMAC_VV) CASE(5, F_FMAC_SV)
    CASE(6, F_FMAAK) CASE(7, F_MUL_VV) CASE(8, F_MUL_SV) CASE(9, F_MUL_LIT) CASE(10, F_ADD_VV) CASE(11, F_SUB_ABS_V)
    CASE(12, F_SUB_ABS_S) CASE(13, F_MAX_VV) CASE(14, F_MAX3) CASE(15, F_MAXIMUM3) CASE(16, F_MAXIMUM3_2)
    CASE(17, F_CMP_VCC) CASE(18, F_CMP_S) CASE(19, F_CND_VCC) CASE(20, F_CMPCND) CASE(21, F_MOV) CASE(22, F_SQRT)
    CASE(23, F_RCP) CASE(26, F_CHAIN) CASE(27, F_MUL_CHAIN) CASE(28, F_VALU_SALU) CASE(29, F_VALU_SALU_S)
    CASE(30, F_AND) CASE(31, F_ADD_I) CASE(32, F_BFE)
    if constexpr (FORM == 24) asm volatile(BODY16(F_PK_MUL) : [p0] "+v"(p0) : [p1] "v"(p1), [p2] "v"(p2));
    if constexpr (FORM == 25) asm volatile(BODY16(F_PK_FMA) : [p0] "+v"(p0) : [p1] "v"(p1), [p2] "v"(p2));
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7)) + p0.x + p0.y;
}

static int g_only = -1;   // argv[1]: run one form only

template <int FORM>
void run(const char* name, float* out, int valu_per_iter = 16) {
  if (g_only >= 0 && g_only != FORM) return;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("%-46s", name); fflush(stdout);
  for (int wps : {1, 2, 5, 8}) {               // waves per SIMD: blocks of 4 waves, 256 CUs
    int blocks = 256 * wps;
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0); k<FORM><<<blocks, 256>>>(out, 0.999f, 1e-3f); (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double cycles = best * 1e-3 * 2.4e9;                      // SIMD cycles at the nominal 2.4 GHz
    double per_simd = (double)ITER * valu_per_iter * wps;     // VALU wave-instructions per SIMD
    printf("  %dw %5.2f", wps, cycles / per_simd);
  }
  printf("\n"); fflush(stdout);
}

int main(int argc, char** argv) {
  if (argc > 1) g_only = atoi(argv[1]);
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  if (g_only < 0) printf("cycles per VALU wave-instruction per SIMD, by waves per SIMD\n");
  run<7>("v_mul_f32 v,v            (VOP2)", out);
  run<8>("v_mul_f32 s,v            (VOP2, SGPR src)", out);
  run<9>("v_mul_f32 literal,v      (VOP2)", out);
  run<10>("v_add_f32 v,v            (VOP2)", out);
  run<13>("v_max_f32 v,v            (VOP2)", out);
  run<30>("v_and_b32 literal,v      (VOP2)", out);
  run<31>("v_add_u32 1,v            (VOP2)", out);
  run<32>("v_add_u32 s,v            (VOP2, SGPR src)", out);
  run<21>("v_mov_b32 v              (VOP1)", out);
  run<4>("v_fmac_f32 v,v           (VOP2, d+=a*b)", out);
  run<5>("v_fmac_f32 s,v           (VOP2, SGPR src)", out);
  run<6>("v_fmaak_f32 v,v,literal  (VOP2)", out);
  run<0>("v_fma_f32 v,v,v          (VOP3, 3 VGPRs)", out);
  run<1>("v_fma_f32 a,a,v          (VOP3, 2 VGPRs)", out);
  run<2>("v_fma_f32 v,s,v          (VOP3, SGPR src)", out);
  run<3>("v_fma_f32 v,2.0,v        (VOP3, inline const)", out);
  run<11>("v_sub_f32 |v|,v          (VOP3 modifiers)", out);
  run<12>("v_sub_f32 |v|,s          (VOP3, SGPR src)", out);
  run<14>("v_max3_f32 v,v,v", out);
  run<15>("v_maximum3_f32 v,v,v", out);
  run<16>("v_maximum3_f32 v,a,a     (2 VGPRs)", out);
  run<17>("v_cmp_gt_f32 vcc", out);
  run<18>("v_cmp_gt_f32 s[20:21]", out);
  run<19>("v_cndmask_b32 vcc", out);
  run<20>("v_cmp vcc + v_cndmask vcc (per instr)", out, 32);
  run<22>("v_sqrt_f32", out);
  run<23>("v_rcp_f32", out);
  run<24>("v_pk_mul_f32 (2 flops-lanes per instr)", out);
  run<25>("v_pk_fma_f32", out);
  run<26>("v_fma_f32 one dependent chain", out);
  run<27>("v_mul_f32 one dependent chain", out);
  run<28>("v_mul_f32 v,v + s_add_u32 (per VALU)", out);
  run<29>("v_mul_f32 s,v + s_add_u32 (per VALU)", out);
  return 0;
}

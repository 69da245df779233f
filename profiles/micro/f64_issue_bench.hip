// Micro-benchmark: fp64 VALU cost on MI355X (gfx950) for the instruction forms of the fp64-path exp / log
// (csrc/rm_math.h): SIMD cycles per wave-instruction at 1, 2, 5 waves per SIMD, independent and dependent.
//   hipcc --offload-arch=gfx950 -O3 f64_issue_bench.hip -o f64_issue_bench && ./f64_issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>

#define ITER 2048
#define B8(I) I("%0", "%1") I("%1", "%2") I("%2", "%3") I("%3", "%4") I("%4", "%5") I("%5", "%6") I("%6", "%7") I("%7", "%0")
#define B16(I) B8(I) B8(I)
#define OPS : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c), "v"(d)

#define D_FMA(D, A)    "v_fma_f64 " D ", " D ", %8, %9\n"
#define D_MUL(D, A)    "v_mul_f64 " D ", " D ", %8\n"
#define D_ADD(D, A)    "v_add_f64 " D ", " D ", %9\n"
#define D_RNDNE(D, A)  "v_rndne_f64 " D ", " D "\n"
#define D_LDEXP(D, A)  "v_ldexp_f64 " D ", " D ", 1\n"
#define D_RCP(D, A)    "v_rcp_f64 " D ", " D "\n"
#define D_CHAIN(D, A)  "v_fma_f64 %0, %0, %8, %9\n"
#define D_MULCH(D, A)  "v_mul_f64 %0, %0, %8\n"

template <int FORM>
__global__ void __launch_bounds__(256) k(double* out, float* outf) {
  double v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  double c = 0.999 + 1e-6 * threadIdx.x, d = 1e-3;
  float f0 = threadIdx.x, f1 = f0 + 1.5f;
  for (int i = 0; i < ITER; ++i) {
#define CASE(N, I) if constexpr (FORM == N) asm volatile(B16(I) OPS);
    CASE(0, D_FMA) CASE(1, D_MUL) CASE(2, D_ADD) CASE(3, D_RNDNE) CASE(4, D_LDEXP) CASE(5, D_RCP) CASE(6, D_CHAIN) CASE(7, D_MULCH)
    if constexpr (FORM == 8) {      // 16 conversions f32 -> f64 -> f32 (2 instructions each)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        double t; asm volatile("v_cvt_f64_f32 %0, %1\n" : "=v"(t) : "v"(f0)); asm volatile("v_cvt_f32_f64 %0, %1\n" : "=v"(f0) : "v"(t));
      }
    }
    if constexpr (FORM == 9) {      // 16 IEEE divisions (compiler expansion), two dependent chains
#pragma unroll
      for (int j = 0; j < 8; ++j) { v0 = c / (2.0 + v0); v1 = c / (2.0 + v1); }
    }
    if constexpr (FORM == 10) {     // 16 IEEE fp32 divisions, two chains
#pragma unroll
      for (int j = 0; j < 8; ++j) { f0 = 0.999f / (2.0f + f0); f1 = 0.999f / (2.0f + f1); }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ((v0 + v1) + (v2 + v3)) + ((v4 + v5) + (v6 + v7));
  outf[blockIdx.x * blockDim.x + threadIdx.x] = f0 + f1;
}

template <int FORM>
void run(const char* name, double* out, float* outf, int per_iter = 16) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("%-58s", name); fflush(stdout);
  for (int wps : {1, 2, 5}) {
    int blocks = 256 * wps;
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0); k<FORM><<<blocks, 256>>>(out, outf); (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double cycles = best * 1e-3 * 2.4e9;
    printf("  %dw %6.2f", wps, cycles / ((double)ITER * per_iter * wps));
  }
  printf("\n"); fflush(stdout);
}

int main() {
  double* out; float* outf;
  (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(double)); (void)hipMalloc(&outf, 256 * 8 * 256 * sizeof(float));
  printf("SIMD cycles (2.4 GHz nominal) per wave-instruction, by waves per SIMD\n");
  run<0>("v_fma_f64 independent (8 registers)", out, outf);
  run<1>("v_mul_f64 independent", out, outf);
  run<2>("v_add_f64 independent", out, outf);
  run<6>("v_fma_f64 one dependent chain", out, outf);
  run<7>("v_mul_f64 one dependent chain", out, outf);
  run<3>("v_rndne_f64", out, outf);
  run<4>("v_ldexp_f64", out, outf);
  run<5>("v_rcp_f64", out, outf);
  run<8>("v_cvt_f64_f32 + v_cvt_f32_f64 (per instruction)", out, outf, 16);
  run<9>("IEEE fp64 division c/(2+x), 2 chains (per division)", out, outf);
  run<10>("IEEE fp32 division, 2 chains (per division)", out, outf);
  return 0;
}

// rm_math.h -- the four transcendental functions on the path, written so that the device result is a
// fixed, host-computable function of the input bits (tests sweep all 2^32 inputs, tests/test_math_sweep.py).
//
// What the reference's ATen CPU op stream executes (measured in this repo, oracle/gen_math_golden.py):
//   * x.pow(1/2.33)  (shader.py:37,54,88)   -> Sleef_powf16_u10   (Vectorized<float>::pow)
//   * torch.atan2    (shader.py:99)         -> Sleef_atan2f16_u10 (Vectorized<float>::atan2)
//     Sleef is open source (boost licence; the copy vendored by PyTorch 2.10).  rm_pow / rm_atan2 restate its
//     published algorithm (double-float arithmetic with FMA, AVX-512 getexp/getmant range reduction) operation
//     by operation: every fp32 add, mul, fma and division below is IEEE, so the bits are Sleef's bits.
//   * .log() / .logsumexp() (shader.py:31,49; transformations.py:70) -> MKL VML vmsLn / vmsExp, mode HA
//     (aten/src/ATen/cpu/vml.h).  Closed source, and CPU-dispatched: the same torch build returns different
//     bits on the Intel build container and on the GPU box's AMD EPYC host (profiles/host_math_probe.py), so
//     "the reference's exp/log bits" do not exist as a single target.  rm_exp / rm_log therefore compute the
//     value both MKL variants approximate -- the correctly rounded result -- through fp64 arithmetic
//     (error before the final rounding < 2^-41: the result is the correctly rounded one except on ~1e-5 of inputs;
//     an fp64 FMA issues in ~5.5 cycles per wave on gfx950, profiles/micro/f64_issue_bench.hip).  Against the
//     build container's torch: exp differs on 1.5 % of inputs, log on 0.01 %, never by more than 1 ulp;
//     exhaustive per-block counts in tests/golden/math_sweep.json.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rm {

#define RM_MDEV __device__ __forceinline__

RM_MDEV int f2i(float f) { return __builtin_bit_cast(int, f); }
RM_MDEV float i2f(int i) { return __builtin_bit_cast(float, i); }

// ---------------------------------------------------------------------------
// Correctly rounded square root (= IEEE sqrtf = what ATen's vector_norm ends in): v_sqrt_f32 is 1 ulp, so test
// the two neighbours with exact FMA residuals and step to the one that brackets x (the correction step of LLVM's
// own f32 sqrt expansion; hipcc lowers __fsqrt_rn to the bare v_sqrt_f32 -- NOT correctly rounded, measured).
// 0, inf, NaN and negatives fall through unchanged because every comparison with a NaN residual is false.  The
// input is not pre-scaled, so below 2^-96 (|p| < 1e-14, never on the path) the residuals underflow and the result
// may keep v_sqrt's 1-ulp error; 9 instructions instead of 15.  Exhaustively equal to sqrtf for every x >= 2^-95
// (tests/test_math_sweep.py).  NB the shaders' brightness .pow(1/2) (shader.py:116) is MKL VML vsSqrt in ATen,
// which is NOT correctly rounded (0.6 % of inputs 1 ulp off, host dependent): same situation as exp / log.
// ---------------------------------------------------------------------------
RM_MDEV float sqrt_rn(float x) {
  float r = __builtin_amdgcn_sqrtf(x);
  float lo = i2f(f2i(r) - 1);
  float hi = i2f(f2i(r) + 1);
  float elo = __builtin_fmaf(-lo, r, x);
  float ehi = __builtin_fmaf(-hi, r, x);
  r = (elo <= 0.0f) ? lo : r;
  r = (ehi > 0.0f) ? hi : r;
  return r;
}

// ---------------------------------------------------------------------------
// exp / log through fp64
// ---------------------------------------------------------------------------
RM_MDEV float exp_f64path(float x) {
  const double xd = (double)x;
  const double k = __builtin_rint(xd * 0x1.71547652b82fep+0);
  double r = __builtin_fma(k, -0x1.62e42fefa0000p-1, xd);   // ln2 split: k * hi is exact
  r = __builtin_fma(k, -0x1.cf79abc9e3b3ap-40, r);          // |r| <= 0.3466
  double p = 0x1.27e4fb7789f5cp-22;                          // 1/10!  (Taylor: remainder r^11/11! < 2^-42)
  p = __builtin_fma(p, r, 0x1.71de3a556c734p-19);            // 1/9!
  p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-16);            // 1/8!
  p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-13);            // 1/7!
  p = __builtin_fma(p, r, 0x1.6c16c16c16c17p-10);            // 1/6!
  p = __builtin_fma(p, r, 0x1.1111111111111p-7);             // 1/5!
  p = __builtin_fma(p, r, 0x1.5555555555555p-5);             // 1/4!
  p = __builtin_fma(p, r, 0x1.5555555555555p-3);             // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  float out = (float)__builtin_ldexp(p, (int)k);             // one rounding, subnormal results included
  // outside [-104, 89] the lines above produce garbage (k saturates, r is meaningless): the results there
  // are 0 and +inf; NaN fails both comparisons and has already propagated through p
  out = (x < -104.0f) ? 0.0f : out;
  out = (x > 89.0f) ? __builtin_inff() : out;
  return out;
}

RM_MDEV float log_f64path(float x) {
  const double xd = (double)x;                               // fp32 subnormals are normal doubles
  const long long b = __builtin_bit_cast(long long, xd);
  int hi = (int)(b >> 32);
  int e = (hi >> 20) - 1023;
  hi = (hi & 0x000fffff) | 0x3ff00000;
  const bool up = hi >= 0x3ff6a09f;                          // mantissa above sqrt(2): halve it
  hi = up ? hi - 0x00100000 : hi;
  e = up ? e + 1 : e;
  const double m = __builtin_bit_cast(double, ((long long)hi << 32) | (b & 0xffffffffll));   // [0.7071, 1.4142)
  const double f = m - 1.0;
  const double s = f / (2.0 + f);                            // IEEE division; log m = 2 atanh(s), |s| <= 0.1716
  const double z = s * s;
  double q = 0x1.e1e1e1e1e1e1ep-5;                           // 1/17  (next term z^9/19 < 2^-50)
  q = __builtin_fma(q, z, 0x1.1111111111111p-4);             // 1/15
  q = __builtin_fma(q, z, 0x1.3b13b13b13b14p-4);             // 1/13
  q = __builtin_fma(q, z, 0x1.745d1745d1746p-4);             // 1/11
  q = __builtin_fma(q, z, 0x1.c71c71c71c71cp-4);             // 1/9
  q = __builtin_fma(q, z, 0x1.2492492492492p-3);             // 1/7
  q = __builtin_fma(q, z, 0x1.999999999999ap-3);             // 1/5
  q = __builtin_fma(q, z, 0x1.5555555555555p-2);             // 1/3
  const double t = s + s;
  const double lm = __builtin_fma(t * z, q, t);
  float out = (float)__builtin_fma((double)e, 0x1.62e42fefa39efp-1, lm);
  out = (x == 0.0f) ? -__builtin_inff() : out;
  out = (x < 0.0f) ? __builtin_nanf("") : out;
  out = (x == __builtin_inff()) ? x : out;
  out = (x != x) ? x : out;
  return out;
}

// ---------------------------------------------------------------------------
// Sleef double-float helpers (FMA forms of sleef/src/common/df.h)
// ---------------------------------------------------------------------------
struct F2 {
  float x, y;
};
RM_MDEV F2 mk2(float x, float y) { return F2{x, y}; }
RM_MDEV float mulsign(float x, float y) { return i2f(f2i(x) ^ (f2i(y) & (int)0x80000000)); }
RM_MDEV F2 df_normalize(F2 t) { float s = t.x + t.y; return mk2(s, (t.x - s) + t.y); }
RM_MDEV F2 df_scale(F2 d, float s) { return mk2(d.x * s, d.y * s); }
RM_MDEV F2 df_neg(F2 d) { return mk2(-d.x, -d.y); }
RM_MDEV F2 df_add_f_f(float x, float y) { float s = x + y; return mk2(s, (x - s) + y); }
RM_MDEV F2 df_add2_f_f(float x, float y) { float s = x + y; float v = s - x; return mk2(s, (x - (s - v)) + (y - v)); }
RM_MDEV F2 df_add_f_f2(float x, F2 y) { float s = x + y.x; return mk2(s, ((x - s) + y.x) + y.y); }
RM_MDEV F2 df_add2_f2_f(F2 x, float y) {
  float s = x.x + y; float v = s - x.x; float t = (x.x - (s - v)) + (y - v);
  return mk2(s, t + x.y);
}
RM_MDEV F2 df_add_f2_f2(F2 x, F2 y) { float s = x.x + y.x; return mk2(s, (((x.x - s) + y.x) + x.y) + y.y); }
RM_MDEV F2 df_add2_f2_f2(F2 x, F2 y) {
  float s = x.x + y.x; float v = s - x.x; float t = (x.x - (s - v)) + (y.x - v);
  return mk2(s, t + (x.y + y.y));
}
RM_MDEV F2 df_mul_f2_f(F2 x, float y) {
  float s = x.x * y;
  return mk2(s, __builtin_fmaf(x.y, y, __builtin_fmaf(x.x, y, -s)));
}
RM_MDEV F2 df_mul_f2_f2(F2 x, F2 y) {
  float s = x.x * y.x;
  return mk2(s, __builtin_fmaf(x.x, y.y, __builtin_fmaf(x.y, y.x, __builtin_fmaf(x.x, y.x, -s))));
}
RM_MDEV F2 df_squ(F2 x) {
  float s = x.x * x.x;
  return mk2(s, __builtin_fmaf(x.x + x.x, x.y, __builtin_fmaf(x.x, x.x, -s)));
}
RM_MDEV F2 df_div(F2 n, F2 d) {
  float t = 1.0f / d.x;                                       // IEEE division (vrec_vf_vf = div on AVX-512)
  float sx = n.x * t;
  float u = __builtin_fmaf(t, n.x, -sx);
  float v = __builtin_fmaf(-d.y, t, __builtin_fmaf(-d.x, t, 1.0f));
  return mk2(sx, __builtin_fmaf(sx, v, __builtin_fmaf(n.y, t, u)));
}

// vgetexpps / vgetmantps(_MM_MANT_NORM_p75_1p5, _MM_MANT_SIGN_nan) on a non-negative argument
RM_MDEV float getexp_pos(float d) {          // floor(log2 d); 0 -> -inf, inf -> inf, NaN -> NaN
  if (d != d) return d;
  if (d == 0.0f) return -__builtin_inff();
  if (d == __builtin_inff()) return d;
  int b = f2i(d), ex = (b >> 23) & 0xff;
  if (ex == 0) {                             // subnormal: position of the leading mantissa bit
    int lz = __builtin_clz((unsigned)(b & 0x7fffff)) - 8;   // 1..23
    return (float)(-126 - lz);
  }
  return (float)(ex - 127);
}
RM_MDEV float getmant_p75_1p5(float d) {     // d >= 0 (or NaN)
  if (d != d) return d;
  if (d == 0.0f || d == __builtin_inff()) return 1.0f;
  int b = f2i(d), ex = (b >> 23) & 0xff, mant = b & 0x7fffff;
  if (ex == 0) {
    int lz = __builtin_clz((unsigned)mant) - 8;
    mant = (mant << lz) & 0x7fffff;
  }
  float m = i2f(0x3f800000 | mant);          // [1, 2)
  return (m >= 1.5f) ? m * 0.5f : m;
}

// logkf (sleefsimdsp.c, ENABLE_AVX512F branch): log(d) as a double-float, d >= 0
RM_MDEV F2 sleef_logkf(float d) {
  float e = getexp_pos(d * (1.0f / 0.75f));
  e = (e == __builtin_inff()) ? 128.0f : e;
  const float m = getmant_p75_1p5(d);
  const F2 x = df_div(df_add2_f_f(-1.0f, m), df_add2_f_f(1.0f, m));
  const F2 x2 = df_squ(x);
  float t = 0.240320354700088500976562f;
  t = __builtin_fmaf(t, x2.x, 0.285112679004669189453125f);
  t = __builtin_fmaf(t, x2.x, 0.400007992982864379882812f);
  const F2 c = mk2(0.66666662693023681640625f, 3.69183861259614332084311e-09f);
  F2 s = df_mul_f2_f(mk2(0.69314718246459960938f, -1.904654323148236017e-09f), e);
  s = df_add_f2_f2(s, df_scale(x, 2.0f));
  s = df_add_f2_f2(s, df_mul_f2_f2(df_mul_f2_f2(x2, x), df_add2_f2_f2(df_mul_f2_f(x2, t), c)));
  return s;
}

RM_MDEV float sleef_ldexpf(float x, int q) {   // vldexp_vf_vf_vi2
  int m = q >> 31;
  m = (((m + q) >> 6) - m) << 4;
  q = q - (m << 2);
  m += 0x7f;
  m = m < 0 ? 0 : m;
  m = m > 0xff ? 0xff : m;
  float u = i2f(m << 23);
  x = x * u * u * u * u;
  u = i2f((q + 0x7f) << 23);
  return x * u;
}

// expkf: exp of a double-float
RM_MDEV float sleef_expkf(F2 d) {
  float u = (d.x + d.y) * 1.442695040888963407359924681001892137426645954152985934135449406931f;
  const int q = (int)__builtin_rintf(u);
  F2 s = df_add2_f2_f(d, (float)q * -0.693145751953125f);
  s = df_add2_f2_f(s, (float)q * -1.428606765330187045e-06f);
  s = df_normalize(s);
  u = 0.00136324646882712841033936f;
  u = __builtin_fmaf(u, s.x, 0.00836596917361021041870117f);
  u = __builtin_fmaf(u, s.x, 0.0416710823774337768554688f);
  u = __builtin_fmaf(u, s.x, 0.166665524244308471679688f);
  u = __builtin_fmaf(u, s.x, 0.499999850988388061523438f);
  F2 t = df_add_f2_f2(s, df_mul_f2_f(df_squ(s), u));
  t = df_add_f_f2(1.0f, t);
  u = sleef_ldexpf(t.x + t.y, q);
  return (d.x < -104.0f) ? 0.0f : u;
}

// Sleef_powf_u10 (xpowf)
RM_MDEV float sleef_powf(float x, float y) {
  const bool yisint = (__builtin_truncf(y) == y) || (fabsf(y) > (float)(1 << 24));
  const bool yisodd = ((1 & (int)y) == 1) && yisint && (fabsf(y) < (float)(1 << 24));
  float result = sleef_expkf(df_mul_f2_f(sleef_logkf(fabsf(x)), y));
  result = (result != result) ? __builtin_inff() : result;
  result *= (x > 0.0f) ? 1.0f : (yisint ? (yisodd ? -1.0f : 1.0f) : __builtin_nanf(""));
  const float efx = mulsign(fabsf(x) - 1.0f, y);
  if (fabsf(y) == __builtin_inff()) result = (efx < 0.0f) ? 0.0f : ((efx == 0.0f) ? 1.0f : __builtin_inff());
  if (fabsf(x) == __builtin_inff() || x == 0.0f) {
    const float v = ((f2i(y) < 0) != (x == 0.0f)) ? 0.0f : __builtin_inff();
    result = mulsign(v, yisodd ? x : 1.0f);
  }
  if (x != x || y != y) result = __builtin_nanf("");
  if (y == 0.0f || x == 1.0f) result = 1.0f;
  return result;
}

// atan2kf_u1 / Sleef_atan2f_u10 (xatan2f_u1)
RM_MDEV F2 sleef_atan2kf_u1(F2 y, F2 x) {
  int q = (x.x < 0.0f) ? -2 : 0;
  if (x.x < 0.0f) { x.x = -x.x; x.y = -x.y; }
  const bool p = x.x < y.x;
  q = p ? q + 1 : q;
  F2 s = p ? df_neg(x) : y;
  F2 t = p ? y : x;
  s = df_div(s, t);
  t = df_squ(s);
  t = df_normalize(t);
  float u = -0.00176397908944636583328247f;
  u = __builtin_fmaf(u, t.x, 0.0107900900766253471374512f);
  u = __builtin_fmaf(u, t.x, -0.0309564601629972457885742f);
  u = __builtin_fmaf(u, t.x, 0.0577365085482597351074219f);
  u = __builtin_fmaf(u, t.x, -0.0838950723409652709960938f);
  u = __builtin_fmaf(u, t.x, 0.109463557600975036621094f);
  u = __builtin_fmaf(u, t.x, -0.142626821994781494140625f);
  u = __builtin_fmaf(u, t.x, 0.199983194470405578613281f);
  t = df_mul_f2_f2(t, df_add_f_f(-0.333332866430282592773438f, u * t.x));
  t = df_mul_f2_f2(s, df_add_f_f2(1.0f, t));
  t = df_add_f2_f2(df_mul_f2_f(mk2(1.5707963705062866211f, -4.3711388286737928865e-08f), (float)q), t);
  return t;
}
RM_MDEV float sleef_isinf2(float d, float m) {
  return (fabsf(d) == __builtin_inff()) ? i2f((f2i(d) & (int)0x80000000) | f2i(m)) : 0.0f;
}
RM_MDEV float sleef_atan2f(float y, float x) {
  if (fabsf(x) < 2.9387372783541830947e-39f) { x *= (float)(1 << 24); y *= (float)(1 << 24); }
  const F2 d = sleef_atan2kf_u1(mk2(fabsf(y), 0.0f), mk2(x, 0.0f));
  float r = d.x + d.y;
  r = mulsign(r, x);
  const float pi2 = 1.57079637050628662109375f, pi4 = 0.785398185253143310546875f;
  if (fabsf(x) == __builtin_inff() || x == 0.0f) r = pi2 - sleef_isinf2(x, mulsign(pi2, x));
  if (fabsf(y) == __builtin_inff()) r = pi2 - sleef_isinf2(x, mulsign(pi4, x));
  if (y == 0.0f) r = (f2i(x) < 0) ? 3.1415927410125732421875f : 0.0f;
  r = mulsign(r, y);
  return (x != x || y != y) ? __builtin_nanf("") : r;
}

}  // namespace rm

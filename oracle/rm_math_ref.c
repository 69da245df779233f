/* rm_math_ref.c -- CPU restatement of the four transcendental functions on the path.
 * TEST INFRASTRUCTURE ONLY: built into oracle/_build/librm_math_ref.so by oracle/Makefile and loaded by
 * tests/ and oracle/gen_math_golden.py; the product (ray_marching_amd/) never links or loads it.
 *
 * What the reference executes for these calls is ATen's CPU kernels of PyTorch 2.10 (not under
 * /root/reference); measured in this repo (oracle/gen_math_golden.py, profiles/host_math_probe.py):
 *   x.pow(1/2.33)   shader.py:37,54,88   Sleef_powf16_u10   (sleef 3.x vendored by PyTorch, src/libm/sleefsimdsp.c xpowf)
 *   torch.atan2     shader.py:99         Sleef_atan2f16_u10 (xatan2f_u1)
 *   .log() .logsumexp()  shader.py:31,49, transformations.py:70   MKL VML vmsLn / vmsExp, VML_HA (closed source;
 *                   results depend on the host CPU) -> ref_expf / ref_logf are the fp64-path functions the
 *                   device uses (csrc/rm_math.h), and the sweep records how often torch on THIS host differs.
 * Parity pin: ref_powf / ref_atan2f equal torch on every input tried (all 2^32 x for the shader's exponent,
 * 2^32 hashed (y,x) pairs); tests/golden/math_sweep.json holds the per-block checksums and mismatch counts.
 *
 * Plain C, every operation IEEE (compile with -ffp-contract=off -mfma; fma()/fmaf() are the only fused ops).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline int32_t fbits(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
static inline float bitsf(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

/* ---- exp / log through double precision ----------------------------------------------------------- */
float ref_expf(float x) {
  if (x != x) return x;
  if (x < -104.0f) return 0.0f;
  if (x > 89.0f) return INFINITY;
  double xd = (double)x;
  double k = rint(xd * 0x1.71547652b82fep+0);
  double r = fma(k, -0x1.62e42fefa0000p-1, xd);
  r = fma(k, -0x1.cf79abc9e3b3ap-40, r);
  static const double c[] = {0x1.27e4fb7789f5cp-22, 0x1.71de3a556c734p-19,
                             0x1.a01a01a01a01ap-16, 0x1.a01a01a01a01ap-13, 0x1.6c16c16c16c17p-10,
                             0x1.1111111111111p-7,  0x1.5555555555555p-5,  0x1.5555555555555p-3,
                             0.5, 1.0, 1.0};
  double p = c[0];
  for (int i = 1; i < 11; ++i) p = fma(p, r, c[i]);
  return (float)ldexp(p, (int)k);
}

float ref_logf(float x) {
  if (x != x) return x;
  if (x == 0.0f) return -INFINITY;
  if (x < 0.0f) return NAN;
  if (isinf(x)) return x;
  double xd = (double)x;
  int64_t b; memcpy(&b, &xd, 8);
  int32_t hi = (int32_t)(b >> 32);
  int e = (hi >> 20) - 1023;
  hi = (hi & 0x000fffff) | 0x3ff00000;
  if (hi >= 0x3ff6a09f) { hi -= 0x00100000; e += 1; }
  int64_t mb = ((int64_t)hi << 32) | (b & 0xffffffffll);
  double m; memcpy(&m, &mb, 8);
  double f = m - 1.0;
  double s = f / (2.0 + f);
  double z = s * s;
  static const double c[] = {0x1.e1e1e1e1e1e1ep-5, 0x1.1111111111111p-4, 0x1.3b13b13b13b14p-4, 0x1.745d1745d1746p-4,
                             0x1.c71c71c71c71cp-4, 0x1.2492492492492p-3, 0x1.999999999999ap-3, 0x1.5555555555555p-2};
  double q = c[0];
  for (int i = 1; i < 8; ++i) q = fma(q, z, c[i]);
  double t = s + s;
  double lm = fma(t * z, q, t);
  return (float)fma((double)e, 0x1.62e42fefa39efp-1, lm);
}

/* ---- Sleef double-float arithmetic, FMA forms (sleef/src/common/df.h) ------------------------------ */
typedef struct { float x, y; } df;
static inline df mk(float x, float y) { df r = {x, y}; return r; }
static inline float mulsign(float x, float y) { return bitsf(fbits(x) ^ (fbits(y) & (int32_t)0x80000000)); }
static inline df dfnormalize(df t) { float s = t.x + t.y; return mk(s, (t.x - s) + t.y); }
static inline df dfscale(df d, float s) { return mk(d.x * s, d.y * s); }
static inline df dfneg(df d) { return mk(-d.x, -d.y); }
static inline df dfadd_f_f(float x, float y) { float s = x + y; return mk(s, (x - s) + y); }
static inline df dfadd2_f_f(float x, float y) { float s = x + y, v = s - x; return mk(s, (x - (s - v)) + (y - v)); }
static inline df dfadd_f_df(float x, df y) { float s = x + y.x; return mk(s, ((x - s) + y.x) + y.y); }
static inline df dfadd2_df_f(df x, float y) { float s = x.x + y, v = s - x.x, t = (x.x - (s - v)) + (y - v); return mk(s, t + x.y); }
static inline df dfadd_df_df(df x, df y) { float s = x.x + y.x; return mk(s, (((x.x - s) + y.x) + x.y) + y.y); }
static inline df dfadd2_df_df(df x, df y) { float s = x.x + y.x, v = s - x.x, t = (x.x - (s - v)) + (y.x - v); return mk(s, t + (x.y + y.y)); }
static inline df dfmul_df_f(df x, float y) { float s = x.x * y; return mk(s, fmaf(x.y, y, fmaf(x.x, y, -s))); }
static inline df dfmul_df_df(df x, df y) { float s = x.x * y.x; return mk(s, fmaf(x.x, y.y, fmaf(x.y, y.x, fmaf(x.x, y.x, -s)))); }
static inline df dfsqu(df x) { float s = x.x * x.x; return mk(s, fmaf(x.x + x.x, x.y, fmaf(x.x, x.x, -s))); }
static inline df dfdiv(df n, df d) {
  float t = 1.0f / d.x, sx = n.x * t, u = fmaf(t, n.x, -sx), v = fmaf(-d.y, t, fmaf(-d.x, t, 1.0f));
  return mk(sx, fmaf(sx, v, fmaf(n.y, t, u)));
}

/* vgetexpps / vgetmantps(_MM_MANT_NORM_p75_1p5, _MM_MANT_SIGN_nan), argument >= 0 */
static float getexp_pos(float d) {
  if (d != d) return d;
  if (d == 0.0f) return -INFINITY;
  if (isinf(d)) return d;
  int e; frexpf(d, &e);
  return (float)(e - 1);
}
static float getmant_p75_1p5(float d) {
  if (d != d) return d;
  if (d == 0.0f || isinf(d)) return 1.0f;
  int e; float m = 2.0f * frexpf(d, &e);   /* [1,2) */
  return (m >= 1.5f) ? 0.5f * m : m;
}

static df logkf(float d) {                 /* sleefsimdsp.c logkf, ENABLE_AVX512F branch */
  float e = getexp_pos(d * (1.0f / 0.75f));
  if (isinf(e) && e > 0) e = 128.0f;
  float m = getmant_p75_1p5(d);
  df x = dfdiv(dfadd2_f_f(-1.0f, m), dfadd2_f_f(1.0f, m));
  df x2 = dfsqu(x);
  float t = 0.240320354700088500976562f;
  t = fmaf(t, x2.x, 0.285112679004669189453125f);
  t = fmaf(t, x2.x, 0.400007992982864379882812f);
  df c = mk(0.66666662693023681640625f, 3.69183861259614332084311e-09f);
  df s = dfmul_df_f(mk(0.69314718246459960938f, -1.904654323148236017e-09f), e);
  s = dfadd_df_df(s, dfscale(x, 2.0f));
  s = dfadd_df_df(s, dfmul_df_df(dfmul_df_df(x2, x), dfadd2_df_df(dfmul_df_f(x2, t), c)));
  return s;
}

static float vldexpf_(float x, int q) {    /* vldexp_vf_vf_vi2 */
  int m = q >> 31;
  m = (((m + q) >> 6) - m) << 4;
  q = q - (m << 2);
  m += 0x7f;
  m = m < 0 ? 0 : m;
  m = m > 0xff ? 0xff : m;
  float u = bitsf(m << 23);
  x = x * u * u * u * u;
  u = bitsf((q + 0x7f) << 23);
  return x * u;
}

static float expkf(df d) {
  float u = (d.x + d.y) * 1.442695040888963407359924681001892137426645954152985934135449406931f;
  int q = (int)rintf(u);
  df s = dfadd2_df_f(d, (float)q * -0.693145751953125f);
  s = dfadd2_df_f(s, (float)q * -1.428606765330187045e-06f);
  s = dfnormalize(s);
  u = 0.00136324646882712841033936f;
  u = fmaf(u, s.x, 0.00836596917361021041870117f);
  u = fmaf(u, s.x, 0.0416710823774337768554688f);
  u = fmaf(u, s.x, 0.166665524244308471679688f);
  u = fmaf(u, s.x, 0.499999850988388061523438f);
  df t = dfadd_df_df(s, dfmul_df_f(dfsqu(s), u));
  t = dfadd_f_df(1.0f, t);
  u = vldexpf_(t.x + t.y, q);
  return (d.x < -104.0f) ? 0.0f : u;
}

float ref_powf(float x, float y) {         /* xpowf */
  int yisint = (truncf(y) == y) || (fabsf(y) > (float)(1 << 24));
  int yisodd = ((1 & (int)y) == 1) && yisint && (fabsf(y) < (float)(1 << 24));
  float result = expkf(dfmul_df_f(logkf(fabsf(x)), y));
  if (result != result) result = INFINITY;
  result *= (x > 0.0f) ? 1.0f : (yisint ? (yisodd ? -1.0f : 1.0f) : NAN);
  float efx = mulsign(fabsf(x) - 1.0f, y);
  if (isinf(y)) result = (efx < 0.0f) ? 0.0f : ((efx == 0.0f) ? 1.0f : INFINITY);
  if (isinf(x) || x == 0.0f) {
    float v = ((fbits(y) < 0) != (x == 0.0f)) ? 0.0f : INFINITY;
    result = mulsign(v, yisodd ? x : 1.0f);
  }
  if (x != x || y != y) result = NAN;
  if (y == 0.0f || x == 1.0f) result = 1.0f;
  return result;
}

static df atan2kf_u1(df y, df x) {
  int q = (x.x < 0.0f) ? -2 : 0;
  if (x.x < 0.0f) { x.x = -x.x; x.y = -x.y; }
  int p = x.x < y.x;
  if (p) q += 1;
  df s = p ? dfneg(x) : y;
  df t = p ? y : x;
  s = dfdiv(s, t);
  t = dfsqu(s);
  t = dfnormalize(t);
  float u = -0.00176397908944636583328247f;
  u = fmaf(u, t.x, 0.0107900900766253471374512f);
  u = fmaf(u, t.x, -0.0309564601629972457885742f);
  u = fmaf(u, t.x, 0.0577365085482597351074219f);
  u = fmaf(u, t.x, -0.0838950723409652709960938f);
  u = fmaf(u, t.x, 0.109463557600975036621094f);
  u = fmaf(u, t.x, -0.142626821994781494140625f);
  u = fmaf(u, t.x, 0.199983194470405578613281f);
  t = dfmul_df_df(t, dfadd_f_f(-0.333332866430282592773438f, u * t.x));
  t = dfmul_df_df(s, dfadd_f_df(1.0f, t));
  t = dfadd_df_df(dfmul_df_f(mk(1.5707963705062866211f, -4.3711388286737928865e-08f), (float)q), t);
  return t;
}
static inline float isinf2(float d, float m) {
  return isinf(d) ? bitsf((fbits(d) & (int32_t)0x80000000) | fbits(m)) : 0.0f;
}
float ref_atan2f(float y, float x) {       /* xatan2f_u1 */
  if (fabsf(x) < 2.9387372783541830947e-39f) { x *= (float)(1 << 24); y *= (float)(1 << 24); }
  df d = atan2kf_u1(mk(fabsf(y), 0.0f), mk(x, 0.0f));
  float r = d.x + d.y;
  r = mulsign(r, x);
  const float pi2 = (float)(M_PI / 2), pi4 = (float)(M_PI / 4);
  if (isinf(x) || x == 0.0f) r = pi2 - isinf2(x, mulsign(pi2, x));
  if (isinf(y)) r = pi2 - isinf2(x, mulsign(pi4, x));
  if (y == 0.0f) r = (fbits(x) < 0) ? (float)M_PI : 0.0f;
  r = mulsign(r, y);
  return (x != x || y != y) ? NAN : r;
}

/* ---- sweep support ---------------------------------------------------------------------------------
 * Input i (0 .. 2^32-1) of a unary sweep is the float with bit pattern i; of the atan2 sweep the pair
 * (y, x) = (bits hash_y(i), bits hash_x(i)).  A block is 2^24 consecutive inputs.  Checksum of a block =
 * sum over its inputs of canon(out_bits) * (2 i + 1) mod 2^64 (NaNs canonicalised to 0x7fc00000). */
enum { REF_EXP = 0, REF_LOG = 1, REF_POW_GAMMA = 2, REF_ATAN2 = 3, REF_SQRT = 4 };

/* IEEE square root.  ATen's vector_norm ends in it; the shaders' brightness .pow(1/2) (shader.py:116) is MKL VML
 * vsSqrt instead, which is not correctly rounded -- the sweep records how often this host's torch.sqrt differs. */
float ref_sqrtf(float x) { return sqrtf(x); }

uint32_t rm_sweep_hash_y(uint32_t i) { uint32_t h = i * 0x9E3779B1u + 0x7F4A7C15u; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; return h; }
uint32_t rm_sweep_hash_x(uint32_t i) { uint32_t h = (i ^ 0x85EBCA6Bu) * 0xC2B2AE35u; h ^= h >> 13; h *= 0x297A2D39u; h ^= h >> 16; return h; }

static inline uint32_t canon(float f) { return (f != f) ? 0x7fc00000u : (uint32_t)fbits(f); }
static inline float sweep_eval(int fn, uint32_t i) {
  switch (fn) {
    case REF_EXP: return ref_expf(bitsf((int32_t)i));
    case REF_LOG: return ref_logf(bitsf((int32_t)i));
    case REF_POW_GAMMA: return ref_powf(bitsf((int32_t)i), (float)(1 / 2.33));
    case REF_SQRT: return ref_sqrtf(bitsf((int32_t)i));
    default: return ref_atan2f(bitsf((int32_t)rm_sweep_hash_y(i)), bitsf((int32_t)rm_sweep_hash_x(i)));
  }
}

/* inputs of one block, for the torch side of the comparison (a: x or y; b: atan2's x) */
void rm_sweep_inputs(int fn, uint32_t block, float* a, float* b) {
  const uint32_t base = block << 24;
#pragma omp parallel for
  for (int64_t j = 0; j < (1 << 24); ++j) {
    uint32_t i = base + (uint32_t)j;
    if (fn == REF_ATAN2) { a[j] = bitsf((int32_t)rm_sweep_hash_y(i)); b[j] = bitsf((int32_t)rm_sweep_hash_x(i)); }
    else a[j] = bitsf((int32_t)i);
  }
}

/* Checksum of this file's function over one block; when `other` (another implementation's outputs for the
 * same block, e.g. torch's) is given also its checksum, the number of inputs on which the two differ, and the
 * largest distance in units in the last place (bit patterns compared as ordered integers; NaN == NaN). */
uint64_t rm_sweep_block(int fn, uint32_t block, const float* other, uint64_t* other_sum, int64_t* n_diff, int64_t* max_ulp) {
  const uint32_t base = block << 24;
  uint64_t sum = 0, osum = 0;
  int64_t nd = 0, mu = 0;
#pragma omp parallel for reduction(+ : sum, osum, nd) reduction(max : mu)
  for (int64_t j = 0; j < (1 << 24); ++j) {
    uint32_t i = base + (uint32_t)j;
    uint32_t mine = canon(sweep_eval(fn, i));
    uint64_t w = 2ull * i + 1ull;
    sum += (uint64_t)mine * w;
    if (other) {
      uint32_t o = canon(other[j]);
      osum += (uint64_t)o * w;
      if (o != mine) {
        nd += 1;
        int64_t a = (mine & 0x80000000u) ? -(int64_t)(mine & 0x7fffffffu) : (int64_t)mine;
        int64_t b = (o & 0x80000000u) ? -(int64_t)(o & 0x7fffffffu) : (int64_t)o;
        int64_t d = a > b ? a - b : b - a;
        if (d > mu) mu = d;
      }
    }
  }
  if (other_sum) *other_sum = osum;
  if (n_diff) *n_diff = nd;
  if (max_ulp) *max_ulp = mu;
  return sum;
}

/* plain array forms (tests) */
void ref_expf_v(const float* a, float* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = ref_expf(a[i]); }
void ref_sqrtf_v(const float* a, float* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = ref_sqrtf(a[i]); }
void ref_logf_v(const float* a, float* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = ref_logf(a[i]); }
void ref_powf_v(const float* a, const float* b, float* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = ref_powf(a[i], b[i]); }
void ref_atan2f_v(const float* a, const float* b, float* out, int64_t n) { for (int64_t i = 0; i < n; ++i) out[i] = ref_atan2f(a[i], b[i]); }

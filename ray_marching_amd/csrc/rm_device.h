// rm_device.h -- device-side SDF evaluation for gfx950 (MI355X, wave64).
//
// One implementation of every scene-graph node (forward value and reverse-mode
// VJP), written once as straight-line handlers over a small per-ray machine
// state.  Two drivers run the handlers:
//   * RuntimeProgram : a wave-uniform interpreter loop over a program staged in
//     LDS (any scene, no recompilation);
//   * StaticProgram  : the same handlers unrolled at compile time over a
//     constexpr program (stack/tape become registers, scalar branches vanish).
//
// Arithmetic follows the reference's ATen CPU op stream operation by operation
// (compile with -ffp-contract=off): products and sums round separately except
// inside vector norms, where ATen's kernel is an FMA chain
// fma(z,z,fma(y,y,x*x)) -- measured in this repo, see DESIGN.md "Numerics".
// Comparisons are written so NaNs propagate the way torch's min/max/clamp/where
// do.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rm_abi.h"
#include "rm_math.h"

namespace rm {

// --------------------------------------------------------------------------
// small vector type
// --------------------------------------------------------------------------
struct V3 {
  float x, y, z;
};

#define RM_DEV __device__ __forceinline__

RM_DEV V3 mk3(float x, float y, float z) { return V3{x, y, z}; }
RM_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RM_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RM_DEV V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
RM_DEV V3 neg(V3 a) { return V3{-a.x, -a.y, -a.z}; }
// dot with torch's .mul().sum(-1) rounding: three rounded products, sequential adds.
RM_DEV float dot_seq(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// quaternion.py:18-21 -- each product rounded, then the difference.
RM_DEV V3 cross(V3 a, V3 b) {
  return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// Correctly rounded sqrt (rm_math.h: sqrt_rn); the opt-in fast build takes the bare 1-ulp v_sqrt_f32.
RM_DEV float rm_sqrt(float x) {
#if defined(RM_FAST_MATH)
  return __builtin_amdgcn_sqrtf(x);
#elif defined(RM_LIBM_SQRT)
  return __builtin_sqrtf(x);
#else
  return sqrt_rn(x);
#endif
}
// exp / log / pow / atan2 (rm_math.h).  torch's CPU exp and log are MKL VML (closed source, and the bits depend
// on the host CPU), so the exact build computes the correctly rounded value through fp64; pow and atan2 are
// Sleef's u10 algorithms restated, bit-identical with torch.  The opt-in fast build takes the hardware forms
// (v_exp_f32 / v_log_f32, ~2 ulp) and ocml's powf / atan2f.
RM_DEV float rm_exp(float x) {
#if defined(RM_FAST_MATH)
  return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
#elif defined(RM_AB_OCML_EXPLOG)
  return expf(x);
#else
  return exp_f64path(x);
#endif
}
RM_DEV float rm_log(float x) {
#if defined(RM_FAST_MATH)
  return __builtin_amdgcn_logf(x) * 0.693147180559945309417f;
#elif defined(RM_AB_OCML_EXPLOG)
  return logf(x);
#else
  return log_f64path(x);
#endif
}
RM_DEV float rm_pow(float x, float y) {
#if defined(RM_FAST_MATH) || defined(RM_AB_OCML_SHADER)
  return powf(x, y);
#else
  return sleef_powf(x, y);
#endif
}
RM_DEV float rm_atan2(float y, float x) {
#if defined(RM_FAST_MATH) || defined(RM_AB_OCML_SHADER)
  return atan2f(y, x);
#else
  return sleef_atan2f(y, x);
#endif
}
// sum_{i<n} e(i) with the association ATen's CPU sum kernel uses for a contiguous inner reduction of n fp32
// values (aten/src/ATen/native/cpu/SumKernel.cpp, measured against torch.sum in oracle tests): fewer than 8
// elements -> four interleaved partial sums (row_sum, ilp_factor 4), elements past the last full group of four
// going to the first; 8 or more -> 8-lane vector partial sums, again four interleaved, the scalar tail first
// and the 8 lanes added one after the other.  n < 512 (beyond that the kernel's cascade levels start; the host
// compiler rejects such unions).  torch.logsumexp sums its exp() terms with this kernel (transformations.py:70),
// and a plain left-to-right sum differs from it in the last bit for most inputs once n > 4.
template <class F>
RM_DEV float aten_inner_sum(int n, F e) {
  if (n < 8) {
    if (n < 4) {
      float s = 0.0f;
      for (int i = 0; i < n; ++i) s = s + e(i);
      return s;
    }
    float p0 = e(0), p1 = e(1), p2 = e(2), p3 = e(3);
    for (int i = 4; i < n; ++i) p0 = p0 + e(i);
    return ((p0 + p1) + p2) + p3;
  }
  const int nv = n >> 3, k = nv >> 2;
  float fin = 0.0f;
  for (int i = nv * 8; i < n; ++i) fin = fin + e(i);
  for (int l = 0; l < 8; ++l) {
    float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
    for (int i = 0; i < k; ++i) {
      p0 = p0 + e((4 * i) * 8 + l);
      p1 = p1 + e((4 * i + 1) * 8 + l);
      p2 = p2 + e((4 * i + 2) * 8 + l);
      p3 = p3 + e((4 * i + 3) * 8 + l);
    }
    for (int i = 4 * k; i < nv; ++i) p0 = p0 + e(i * 8 + l);
    fin = fin + (((p0 + p1) + p2) + p3);
  }
  return fin;
}
// The same association with the element count known at compile time and every loop unrolled, so that e(i) is called
// with constant indices: a tape kept in registers (RegStore) must never be indexed dynamically, or the whole store
// goes to scratch memory (measured on the 32-child smooth union of config 5: 1732 B of scratch per lane, 16.7 GB of
// HBM writes per 7680x540 band).
template <int N, class F>
RM_DEV float aten_inner_sum_n(F e) {
  if constexpr (N < 4) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) s = s + e(i);
    return s;
  } else if constexpr (N < 8) {
    float p0 = e(0), p1 = e(1), p2 = e(2), p3 = e(3);
#pragma unroll
    for (int i = 4; i < N; ++i) p0 = p0 + e(i);
    return ((p0 + p1) + p2) + p3;
  } else {
    constexpr int nv = N >> 3, k = nv >> 2;
    float fin = 0.0f;
#pragma unroll
    for (int i = nv * 8; i < N; ++i) fin = fin + e(i);
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#pragma unroll
      for (int i = 0; i < k; ++i) {
        p0 = p0 + e((4 * i) * 8 + l);
        p1 = p1 + e((4 * i + 1) * 8 + l);
        p2 = p2 + e((4 * i + 2) * 8 + l);
        p3 = p3 + e((4 * i + 3) * 8 + l);
      }
#pragma unroll
      for (int i = 4 * k; i < nv; ++i) p0 = p0 + e(i * 8 + l);
      fin = fin + (((p0 + p1) + p2) + p3);
    }
    return fin;
  }
}
// ATen vector_norm (p=2) on CPU: FMA chain then correctly rounded sqrt.
RM_DEV float norm3(V3 a) { return rm_sqrt(__builtin_fmaf(a.z, a.z, __builtin_fmaf(a.y, a.y, a.x * a.x))); }
RM_DEV float norm2(float a, float b) { return rm_sqrt(__builtin_fmaf(b, b, a * a)); }
// Inside a VJP nothing has to be bit-exact (the gradient contract is 1e-4; a VJP's own forward half only feeds
// sub-gradient choices and softmax weights): kFast takes the bare 1-ulp v_sqrt_f32 / v_rcp_f32 in place of the
// 9-instruction exact square root and the ~10-instruction IEEE division.  Values (march, distances, normals, images)
// never come through kFast = true.
#ifndef RM_FAST_VJP
#define RM_FAST_VJP 1
#endif
template <bool kFast> RM_DEV float sqrt_t(float x) { return kFast ? __builtin_amdgcn_sqrtf(x) : rm_sqrt(x); }
template <bool kFast> RM_DEV float norm3_t(V3 a) { return sqrt_t<kFast>(__builtin_fmaf(a.z, a.z, __builtin_fmaf(a.y, a.y, a.x * a.x))); }
template <bool kFast> RM_DEV float norm2_t(float a, float b) { return sqrt_t<kFast>(__builtin_fmaf(b, b, a * a)); }
template <bool kFast> RM_DEV float div_t(float a, float b) { return kFast ? a * __builtin_amdgcn_rcpf(b) : a / b; }
// quaternion.py:55-72: V + w*t + qv x t, t = 2*(qv x V); summed as (y + w*t) + V.
RM_DEV V3 qrot(V3 v, float w, V3 qv) {
  V3 t = 2.0f * cross(qv, v);
  V3 y = cross(qv, t);
  return (y + w * t) + v;
}

// torch semantics helpers ---------------------------------------------------
// gfx950 has NaN-propagating v_minimum3_f32 / v_maximum3_f32 (IEEE-754-2019 minimum/maximum),
// which is exactly torch's min/max/clamp behaviour; v_max_f32 (maxNum) maps NaN to the other
// operand, which is what x.where(x > 0, 0) does with a NaN.
RM_DEV float t_relu_keep(float x) { return __builtin_fmaxf(x, 0.0f); }          // x.where(x > 0, 0)
RM_DEV float t_min(float acc, float d) { return __builtin_elementwise_minimum(acc, d); }  // NaN-propagating
RM_DEV float t_max(float acc, float d) { return __builtin_elementwise_maximum(acc, d); }
RM_DEV float t_clamp(float x, float lo, float hi) { return t_min(t_max(x, lo), hi); }     // NaN passes
RM_DEV float sgn0(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }

RM_DEV float uniform_f(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x)));
}
RM_DEV int uniform_i(int x) { return __builtin_amdgcn_readfirstlane(x); }

// --------------------------------------------------------------------------
// per-ray storage policies (evaluation stack + tape + gradient accumulators)
// --------------------------------------------------------------------------
// LDS columns: element i of this thread lives at base[i * stride]; conflict-free
// because consecutive lanes hit consecutive banks.
struct LdsStore {
  float* base;   // per thread
  int stride;    // wave-uniform; readfirstlane keeps i * stride on the scalar unit (as a VGPR value it
                 // costs a quarter-rate v_mul_lo_u32 per access: measured 368 VALU per interpreted eval)
  // Gradient accumulators (backward kernels): ONE ROW PER WAVE, not a column per thread.  Every accumulation is a wave
  // sum (6 DPP adds in a fixed order) followed by one read-modify-write from lane 63.  Columns cost 65 x 4 B per
  // accumulator and 64-thread block: closed scene 1 (46 accumulators) fitted two waves per CU-quarter, the 32-primitive
  // scene (347) one wave per CU, and 48 primitives did not fit at all; rows cost 4 B per accumulator and wave
  // (profiles/r03_wide_ab.txt: many32 fwd+bwd 29.4 -> 16.7 ms at 256^2, closed scene 1 1.22 -> 1.15 ms at 512^2).
  float* acc_row;   // this wave's row; nullptr in forward kernels
  int acc0;         // store index of accumulator 0
  RM_DEV int at(int i) const { return uniform_i(i) * uniform_i(stride); }
  RM_DEV float ld(int i) const { return base[at(i)]; }
  RM_DEV void st(int i, float v) { base[at(i)] = v; }
  RM_DEV void add(int i, float v);
};

// Registers: with a StaticProgram every index is a compile-time constant after
// inlining, so SROA turns the array into VGPRs.
template <int N>
struct RegStore {
  float r[N > 0 ? N : 1];
  RM_DEV float ld(int i) const { return r[i]; }
  RM_DEV void st(int i, float v) { r[i] = v; }
  RM_DEV void add(int i, float v) { r[i] += v; }
};

// Registers for the evaluation stack and the gradient accumulators, LDS columns for the TAPE: a large smooth union
// keeps one tape slot per child alive across the evaluation of all its children (32 + 1 for the config-5 scene); as
// registers they cost the frame kernel its occupancy (256 VGPRs + 124 AGPRs, 1 wave per SIMD, when the compiler is
// left alone), as scratch memory 16.7 GB of HBM writes per 8K band.  Column stride is a compile-time constant, so
// every access is one ds_read / ds_write with an immediate offset (34 slots x 257 x 4 B < the 64 KiB offset field).
constexpr int kLdsTapeStride = 257;          // blocks of up to 256 threads
template <int NStack, int NSlots, int NTotal>
struct HybridStore {
  float r[(NTotal - NSlots) > 0 ? (NTotal - NSlots) : 1];
  float* base;   // per thread: column of this thread in the block's tape area
  RM_DEV float ld(int i) const {
    if (i < NStack) return r[i];
    if (i < NStack + NSlots) return base[(i - NStack) * kLdsTapeStride];
    return r[i - NSlots];
  }
  RM_DEV void st(int i, float v) {
    if (i < NStack) r[i] = v;
    else if (i < NStack + NSlots) base[(i - NStack) * kLdsTapeStride] = v;
    else r[i - NSlots] = v;
  }
  RM_DEV void add(int i, float v) { st(i, ld(i) + v); }
};

// Parameter block views.  LdsParams reads the staged block where it is needed (generic
// interpreter: the offset is only known at run time).  RegParams copies the block out of LDS
// once per thread into wave-uniform registers (readfirstlane -> SGPRs) so a specialised march
// loop carries no parameter loads at all.
struct LdsParams {
  const float* p;
  RM_DEV float operator[](int i) const { return p[i]; }
  RM_DEV V3 v3(int i) const { return V3{p[i], p[i + 1], p[i + 2]}; }
};

// kVgpr: keep the copies in VGPRs instead (A/B knob, RM_FWD_VGPR_PARAM_LIMIT; off).  An SGPR source operand makes its
// consumer a half-rate instruction (profiles/r03_valu_issue_bench.txt: v_sub |v|,s 4.1 cycles against 2.3 for |v|,v), and
// when the SGPRs are oversubscribed every use of a spilled one is a v_readlane.  Round 3, same-box A/Bs of the config-2
// tile kernel (profiles/r03_ab_vgpr_params.txt, r03_ab_vgpr_final.txt, r03_ab_vs_round2.txt): while ONE instantiation
// of k_render_fwd served inference and recording frames, the SGPR form carried 997 v_readlane and VGPR copies won by 7 %
// (238 -> 221 us; 127.5 -> 108.5 M executed VALU wave-instructions); once the inference instantiation was split off
// (357 v_readlane) the SGPR form is the faster one: 0.205 ms against 0.215 ms with VGPR copies (round-2 tree on that box:
// 0.208).  VGPR copies also cost 22 VGPRs per kernel: the ray pools got 29 % slower with them, the backward kernels
// lose their last wave of occupancy.
template <int N, bool kVgpr = false>
struct RegParams {
  float v[N > 0 ? N : 1];
  RM_DEV void load(const float* lds) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
#if defined(RM_VGPR_PARAMS)   // experiment knob: VGPR-resident copies in every kernel
      v[i] = lds[i];
#else
      v[i] = kVgpr ? lds[i] : uniform_f(lds[i]);
#endif
    }
  }
  RM_DEV float operator[](int i) const { return v[i]; }
  RM_DEV V3 v3(int i) const { return V3{v[i], v[i + 1], v[i + 2]}; }
};

// --------------------------------------------------------------------------
// forward machine
// --------------------------------------------------------------------------
#ifndef RM_CULL_TRACKED
#define RM_CULL_TRACKED 2   // cull sites per scene whose decision is carried from step to step (2 VGPRs each)
#endif

template <class Store, bool Fast = false>
struct Fwd {
  static constexpr bool kFast = Fast;   // forward half of a VJP: 1-ulp square roots (sqrt_t)
  V3 p;        // query point in the current (innermost affine) frame
  float d;     // value register: distance produced by the last node
  float acc;   // running min of the innermost open SDFUnion
  int sp;      // stack pointer (floats)
  int tape0;   // index of tape slot 0 inside the store
  Store* st;
  bool record; // record every fold/onion input (needed by the reverse sweep)
  const float* lds;  // the staged parameter block in LDS (raw + derived), whatever view `P` the handlers read through
  unsigned long long culled;  // wave-uniform: bit s set = the union / smooth-union child folding into tape slot s was skipped
  // Tracked cull sites (StaticProgram, outermost frame, first kCullTracked of them): per-lane bounds
  // lo <= lhs <= hi on the site's test value lhs = slope |p - c| - K at THIS point, carried over from its
  // last full test through the known movement of the point (Scene::eval_near).  NaN = unknown.
  float cull_lo[RM_CULL_TRACKED], cull_hi[RM_CULL_TRACKED];
};
constexpr int kCullTracked = RM_CULL_TRACKED;

// Exact culling of a min-union child (RM_OP_CULL_MIN).  The derived block holds a bounding sphere
// (centre c, radius R; R = +inf when the subtree has none) of the child's surface in the union's frame and
// a slope sigma <= 1, computed from the live parameters at staging time, so child(p) >= sigma |p - c| - R.  If that lower bound --
// taken with a safety margin far above fp32 rounding -- is >= the running minimum `acc` of the children
// evaluated so far for ALL 64 rays, the child cannot lower the minimum and cannot win a tie (ties go to
// the FIRST child), so skipping it leaves value, winner and gradients bit-identical.
template <class S, class PT>
RM_DEV float cull_min_lhs(const S& s, const PT& P, int a0) {
  V3 c = P.v3(a0);
  float K = P[a0 + 3];               // 1.0001 R + 1e-4 + 1e-5 |c|_1, folded at staging (derive_constants)
  float slope = P[a0 + 4];           // sigma - 2e-4; sigma < 1 only under un-normalised affine quaternions
  V3 d = s.p - c;
  float dist = __builtin_amdgcn_sqrtf(__builtin_fmaf(d.z, d.z, __builtin_fmaf(d.y, d.y, d.x * d.x)));
  return __builtin_fmaf(dist, slope, -K);
}
template <class S, class PT>
RM_DEV bool cull_min_test(const S& s, const PT& P, int a0) {
  return __all(cull_min_lhs(s, P, a0) >= s.acc);   // NaN anywhere (K is NaN for an unbounded child): false
}
// The same decision for a tracked site (index T): lhs is 1-Lipschitz in p (slope <= 1), so after the point
// has moved by at most `move` since the bounds were last refreshed, lo - move <= lhs <= hi + move (applied
// by Scene::eval_near).  If lo >= acc for every ray the full test would pass; if hi < acc for some ray it
// would fail; only in between are the 9 instructions of the full test spent, which also refreshes the
// bounds.  Comparisons with NaN (unknown) are false, so an unknown bound always takes the full test.
template <int T, class S, class PT>
RM_DEV bool cull_min_tracked(S& s, const PT& P, int a0) {
  if (__all(s.cull_lo[T] >= s.acc)) return true;
  if (__any(s.cull_hi[T] < s.acc)) return false;
  float lhs = cull_min_lhs(s, P, a0);
  s.cull_lo[T] = lhs; s.cull_hi[T] = lhs;
  return __all(lhs >= s.acc);
}

// Wave-wide max / min of a per-lane float through DPP (no LDS traffic, ~7 VALU instructions): Hillis-Steele inside
// every row of 16 lanes (row_shr 1, 2, 4, 8; lanes without a source keep their own value), then row_bcast:15 and
// row_bcast:31 carry the row results to lane 63.  NaN operands are ignored (IEEE maxNum / minNum).  Needs every lane of
// the wave active -- true wherever a scene is evaluated (lanes past the end of the work run on clamped indices; the
// ray pools give their idle lanes a copy of an active lane's ray).
#define RM_DPP_F(x, ctrl, row_mask) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (x)), __builtin_bit_cast(int, (x)), (ctrl), (row_mask), 0xf, false))
template <bool kMax>
RM_DEV float wave_reduce(float v) {
  auto comb = [](float a, float b) { return kMax ? __builtin_fmaxf(a, b) : __builtin_fminf(a, b); };
  v = comb(v, RM_DPP_F(v, 0x111, 0xf));   // row_shr:1
  v = comb(v, RM_DPP_F(v, 0x112, 0xf));   // row_shr:2
  v = comb(v, RM_DPP_F(v, 0x114, 0xf));   // row_shr:4
  v = comb(v, RM_DPP_F(v, 0x118, 0xf));   // row_shr:8   -> lane 15 of every row holds the row's result
  v = comb(v, RM_DPP_F(v, 0x142, 0xa));   // row_bcast:15 into rows 1 and 3
  v = comb(v, RM_DPP_F(v, 0x143, 0xc));   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's result
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sum over the 64 lanes in a fixed order (the same DPP ladder; lanes without a source add 0); the result is in lane 63.
// Needs every lane active, as above.
#define RM_DPP_Z(x, ctrl, row_mask) \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (x)), (ctrl), (row_mask), 0xf, true))
RM_DEV float wave_sum_lane63(float v) {
  v += RM_DPP_Z(v, 0x111, 0xf);
  v += RM_DPP_Z(v, 0x112, 0xf);
  v += RM_DPP_Z(v, 0x114, 0xf);
  v += RM_DPP_Z(v, 0x118, 0xf);
  v += RM_DPP_Z(v, 0x142, 0xa);
  v += RM_DPP_Z(v, 0x143, 0xc);
  return v;
}
RM_DEV void LdsStore::add(int i, float v) {
  const float sum = wave_sum_lane63(v);
  if ((threadIdx.x & 63) == 63) acc_row[uniform_i(i) - acc0] += sum;
}
// The same accumulator rows on top of a register / hybrid store (specialised backward of scenes with too many
// accumulators for registers: StaticCfg::kRowAcc); `i - acc0` is a compile-time constant after inlining.
template <class Base>
struct RowAccStore : Base {
  float* acc_row;
  int acc0;
  RM_DEV void add(int i, float v) {
    const float sum = wave_sum_lane63(v);
    if ((threadIdx.x & 63) == 63) acc_row[i - acc0] += sum;
  }
};

// Exact culling inside a smooth union (RM_OP_SMOOTH_BEGIN with a bound table, RM_OP_CULL_LSE).
// torch.logsumexp(-k d) = log(sum_i exp(x_i - m)) + m with x_i = -k d_i, m = max x.  exp(x) is EXACTLY +0.0f below
// x = -103.98 (the smallest fp32 subnormal is e^-103.28; rm_math.h: exp_f64path; MKL's exp rounds to zero there too),
// so a child with  k (d_i - d_min) > 104  adds an exact zero to the sum, is not the maximum, and has softmax weight
// exactly 0 in the VJP: leaving it out changes no bit of the value or of any gradient.  Which children those are is
// decided for the whole wave at once, one child per LANE: lane (base + j) reads child j's bounds
//     slope_lb |p - c| - K_lb  <=  child_j(p)  <=  slope_ub |p - c| + K_ub       (subtree_bound, margins folded in)
// and evaluates them on the ball (c0, rho) that holds the points of all 64 rays; d_min <= min_j ub_j, and child j is
// skipped when  k (lb_j - min ub) > 105 (+ the rounding of the products): ~45 instructions per evaluation whatever
// the number of children.  Returns the skip bits at the children's tape slots.
template <class S, class PT>
RM_DEV unsigned long long lse_cull_mask(const S& s, const PT& P, int koff, int table, int base, int n) {
  const int lane = threadIdx.x & 63;
  const float k = P[koff];
  const V3 c0 = mk3(uniform_f(s.p.x), uniform_f(s.p.y), uniform_f(s.p.z));
  const V3 d0 = s.p - c0;
  const float r = __builtin_amdgcn_sqrtf(__builtin_fmaf(d0.z, d0.z, __builtin_fmaf(d0.y, d0.y, d0.x * d0.x)));
  // (a lane whose point is NaN is left out: its value is NaN whichever children are skipped, the nearest one never is)
  const float rho = __builtin_fmaf(wave_reduce<true>(r), 1.0001f, 1e-6f * ((fabsf(c0.x) + fabsf(c0.y)) + fabsf(c0.z)) + 1e-6f);
  const int j = lane - base;
  const bool mine = (j >= 0) & (j < n);
  const float4* e = reinterpret_cast<const float4*>(s.lds + table + 8 * (mine ? j : 0));
  const float4 ea = e[0], eb = e[1];                 // {cx, cy, cz, slope_lb}, {K_lb, slope_ub, K_ub, -}
  const V3 dc = c0 - mk3(ea.x, ea.y, ea.z);
  const float t = __builtin_amdgcn_sqrtf(__builtin_fmaf(dc.z, dc.z, __builtin_fmaf(dc.y, dc.y, dc.x * dc.x)));
  const float lb = __builtin_fmaf(ea.w, __builtin_fmaxf(__builtin_fmaf(t, 0.99999f, -rho), 0.0f), -eb.x);   // K_lb = NaN: unbounded child
  float ub = __builtin_fmaf(eb.y, __builtin_fmaf(t, 1.00001f, rho), eb.z);                                   // +inf: no upper bound known
  ub = mine ? ub : __builtin_inff();
  const float dmin_ub = wave_reduce<false>(ub);
  // x_i - m is formed from rounded products: 2^-23 relative of either term, covered by the last term of the threshold
  const float thr = __builtin_fmaf(1e-6f * fabsf(k), fabsf(lb) + fabsf(dmin_ub), 105.0f);
  const bool cull = mine & (k > 0.0f) & (k * (lb - dmin_ub) > thr);          // NaN anywhere: false
  return __ballot(cull);
}

// Whole-smooth-union culling from the SAME table (RM_OP_CULL_MIN in front of a smooth union, `param offset` field = 1):
//     -lse(-k d) / k  >=  min_j d_j - log(n) / k  >=  min_j lb_j - log(n) / k,
// with the per-child lower bounds lb_j evaluated -- one child per lane -- on the ball that holds the points of all 64
// rays.  ONE sphere around all the children (cull_min_test) is loose for a cluster that fills the room: in the config-5
// scene it lets 21.5 % of the evaluations skip the union, the per-child minimum 30.1 % (profiles/r03_lse_cull_rate.txt).
// Tried only when the one-sphere test has failed.  An unbounded child (K_lb = NaN) makes the minimum -inf.
template <class S, class PT>
RM_DEV bool cull_union_children(const S& s, const PT& P, int koff, int table, int base, int n) {
  const int lane = threadIdx.x & 63;
  const float k = P[koff];
  const V3 c0 = mk3(uniform_f(s.p.x), uniform_f(s.p.y), uniform_f(s.p.z));
  const V3 d0 = s.p - c0;
  const float r = __builtin_amdgcn_sqrtf(__builtin_fmaf(d0.z, d0.z, __builtin_fmaf(d0.y, d0.y, d0.x * d0.x)));
  // a lane whose point is NaN keeps rho finite (fmaxf ignores it) -- and fails the final comparison through its NaN acc
  const float rho = __builtin_fmaf(wave_reduce<true>(r), 1.0001f, 1e-6f * ((fabsf(c0.x) + fabsf(c0.y)) + fabsf(c0.z)) + 1e-6f);
  const int j = lane - base;
  const bool mine = (j >= 0) & (j < n);
  const float4* e = reinterpret_cast<const float4*>(s.lds + table + 8 * (mine ? j : 0));
  const float4 ea = e[0], eb = e[1];
  const V3 dc = c0 - mk3(ea.x, ea.y, ea.z);
  const float t = __builtin_amdgcn_sqrtf(__builtin_fmaf(dc.z, dc.z, __builtin_fmaf(dc.y, dc.y, dc.x * dc.x)));
  float lb = __builtin_fmaf(ea.w, __builtin_fmaxf(__builtin_fmaf(t, 0.99999f, -rho), 0.0f), -eb.x);
  lb = (lb == lb) ? lb : -__builtin_inff();                      // unbounded child: no bound at all
  lb = mine ? lb : __builtin_inff();
  const float lbmin = wave_reduce<false>(lb);
  // log(n) / k with a relative margin, and the absolute one of cull_min_test on top of the children's own (folded into K_lb)
  const float slack = __builtin_fmaf(__builtin_amdgcn_logf((float)n) * 0.693147180559945309417f, 1.001f / k, 1e-4f);
  const float bound = lbmin - slack;
  return (k > 0.0f) && __all(bound >= s.acc);                    // NaN anywhere: false
}

template <class S, class PT>
RM_DEV void fwd_op(S& s, const PT& P, int op, int off, int a0, int a1) {
  switch (op) {
    case RM_OP_SPHERE:  // |p| - r
      s.d = norm3_t<S::kFast>(s.p) - P[off];
      break;
    case RM_OP_BOX: {  // |relu(q)| + min(max(q),0), q = |p| - h
      float qx = fabsf(s.p.x) - P[off], qy = fabsf(s.p.y) - P[off + 1], qz = fabsf(s.p.z) - P[off + 2];
      float m = t_max(t_max(qx, qy), qz);
      // Inside the box every q <= 0, so relu(q) = (+0,+0,+0) and its norm is exactly +0: when that holds
      // for the whole wave (rays inside a room shell: always) the norm -- 3 max, mul, 2 fma and the
      // 9-instruction exact sqrt -- is skipped.  NaN fails (m <= 0) and takes the full path.
      float nr = 0.0f;
      if (!__all(m <= 0.0f)) nr = norm3_t<S::kFast>(mk3(t_relu_keep(qx), t_relu_keep(qy), t_relu_keep(qz)));
      s.d = nr + ((m < 0.0f) ? m : 0.0f);
    } break;
    case RM_OP_PLANE:
      s.d = s.p.x;
      break;
    case RM_OP_LINE: {  // capsule; derived block holds AB and AB/|AB|^2
      V3 ab = P.v3(a0), abs_ = P.v3(a0 + 3);
      V3 ap = s.p - P.v3(off);
      float h = t_clamp(dot_seq(ap, abs_), 0.0f, 1.0f);
      V3 w = mk3(h * ab.x - ap.x, h * ab.y - ap.y, h * ab.z - ap.z);
      s.d = norm3_t<S::kFast>(w) - P[off + 6];
    } break;
    case RM_OP_DISK: {  // axis x, radius in yz
      float rd = norm2_t<S::kFast>(s.p.y, s.p.z) - P[off];
      s.d = norm2_t<S::kFast>(s.p.x, t_relu_keep(rd));
    } break;
    case RM_OP_TORUS: {  // ring in xz
      float ring = norm2_t<S::kFast>(s.p.x, s.p.z) - P[off];
      s.d = norm2_t<S::kFast>(ring, s.p.y) - P[off + 1];
    } break;
    case RM_OP_AFFINE_PUSH: {  // child(rot(p - t, conj(q)))
      s.st->st(s.sp, s.p.x); s.st->st(s.sp + 1, s.p.y); s.st->st(s.sp + 2, s.p.z);
      s.sp += 3;
      V3 t = P.v3(off);
      s.p = qrot(s.p - t, P[off + 3], neg(P.v3(off + 4)));
    } break;
    case RM_OP_AFFINE_POP:
      s.sp -= 3;
      s.p = mk3(s.st->ld(s.sp), s.st->ld(s.sp + 1), s.st->ld(s.sp + 2));
      break;
    case RM_OP_UNION_BEGIN:
      s.st->st(s.sp, s.acc); s.sp += 1;
      s.acc = __builtin_inff();
      break;
    case RM_OP_FOLD_MIN:
      if (s.record) s.st->st(s.tape0 + a0, s.d);
      s.acc = t_min(s.acc, s.d);
      break;
    case RM_OP_UNION_END:
      s.d = s.acc;
      s.sp -= 1; s.acc = s.st->ld(s.sp);
      break;
    case RM_OP_SMOOTH_BEGIN:   // a0 = bound table (0: none), a1 = culling of the children on << 16 | first tape slot << 8 | children
      if (a0 > 0 && (a1 >> 16)) s.culled |= lse_cull_mask(s, P, off, a0, (a1 >> 8) & 255, a1 & 255);
      break;
    case RM_OP_FOLD_LSE:  // children are kept on the tape; the reduction is two-pass like torch.logsumexp
      s.st->st(s.tape0 + a0, s.d);
      break;
    case RM_OP_SMOOTH_END: {  // -logsumexp(-k d)/k : max, sum exp(x - max), log + max, / (-k)
      float nk = -P[off];
      float m = -__builtin_inff();
      // children skipped by their CULL_LSE (bit at their tape slot): exact zeros of the sum, never the maximum
      auto skipped = [&](int i) { return (a0 + i < 64) && ((s.culled >> (a0 + i)) & 1ull); };
      for (int i = 0; i < a1; ++i)
        if (!skipped(i)) m = t_max(m, s.st->ld(s.tape0 + a0 + i) * nk);
      float mm = (fabsf(m) == __builtin_inff()) ? 0.0f : m;
      float L;
      if (s.record) {
        // forward half of a VJP: L only feeds softmax weights and sub-gradient choices of the reverse pass
        // (contract 1e-4), so the hardware exponential / logarithm (~2 ulp) replace the two fp64 paths --
        // ~110 fp64 instructions fewer per smooth union and VJP.  VALUES (march, distances, normals) never
        // come through here: s.record is false for them.
        float sum = 0.0f;
        for (int i = 0; i < a1; ++i)
          if (!skipped(i)) sum = sum + __builtin_amdgcn_exp2f((s.st->ld(s.tape0 + a0 + i) * nk - mm) * 1.44269504088896340736f);
        L = __builtin_amdgcn_logf(sum) * 0.693147180559945309417f + mm;
        s.st->st(s.tape0 + a0 + a1, L);                 // the reverse pass starts from it (slot after the children)
      } else {
        const float sum = aten_inner_sum(a1, [&](int i) {
          return skipped(i) ? 0.0f : rm_exp(s.st->ld(s.tape0 + a0 + i) * nk - mm);
        });
        L = rm_log(sum) + mm;
      }
      s.d = L / nk;
    } break;
    case RM_OP_ROUND:
      s.d = s.d - P[off];
      break;
    case RM_OP_ONION:
      if (s.record) s.st->st(s.tape0 + a0, s.d);
      s.d = fabsf(s.d) - P[off];
      break;
    default:
      break;
  }
}

// SMOOTH_END with compile-time slot range (StaticProgram): the same arithmetic as the case in fwd_op, every tape
// index a constant.  A child whose bit is set in `culled` was skipped by its CULL_LSE: its term is exactly +0.0f and it
// cannot be the maximum (lse_cull_mask), so it is left out of both passes.
template <int A0, int N, bool kCulled, class S, class PT>
RM_DEV void smooth_end_static(S& s, const PT& P, int off) {
  const float nk = -P[off];
  // (kCulled = false -- no bound table on this union's SMOOTH_BEGIN -- leaves straight-line code without a branch per child)
  auto skipped = [&](int i) { return kCulled && (A0 + i < 64) && ((s.culled >> (A0 + i)) & 1ull); };
  float m = -__builtin_inff();
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (!skipped(i)) m = t_max(m, s.st->ld(s.tape0 + A0 + i) * nk);
  const float mm = (fabsf(m) == __builtin_inff()) ? 0.0f : m;
  float L;
  if (s.record) {
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (!skipped(i)) sum = sum + __builtin_amdgcn_exp2f((s.st->ld(s.tape0 + A0 + i) * nk - mm) * 1.44269504088896340736f);
    L = __builtin_amdgcn_logf(sum) * 0.693147180559945309417f + mm;
    s.st->st(s.tape0 + A0 + N, L);
  } else {
    const float sum = aten_inner_sum_n<N>([&](int i) {
      return skipped(i) ? 0.0f : rm_exp(s.st->ld(s.tape0 + A0 + i) * nk - mm);
    });
    L = rm_log(sum) + mm;
  }
  s.d = L / nk;
}

// --------------------------------------------------------------------------
// reverse machine: walks the program backwards.  `g` is dL/d(value register) at
// the current program point, `gp` accumulates dL/d(point) in the current frame.
// Gradient accumulators live in the store at index acc0 + parameter offset
// (raw parameters first, derived constants after them).
// --------------------------------------------------------------------------
template <class Store, bool Acc = true, bool Fast = (RM_FAST_VJP != 0)>
struct Bwd {
  static constexpr bool kAcc = Acc;   // false: point gradient only (no parameter accumulators are touched)
  static constexpr bool kFast = Fast; // 1-ulp square roots and reciprocals (sqrt_t, div_t)
  V3 p;
  V3 gp;
  float g;
  float gframe;  // upstream of the innermost open union / smooth-union
  float fval;    // union: winning slot (as float); smooth-union: logsumexp value L
  int sp;
  int tape0;
  int acc0;
  Store* st;
  unsigned long long culled;   // from the recording forward pass
};

// parameter-gradient accumulation; compiled out of the point-gradient-only pass
template <class S>
RM_DEV void padd(S& s, int i, float v) {
  if constexpr (S::kAcc) s.st->add(i, v);
}

template <class S>
RM_DEV V3 safe_unit_scaled(V3 w, float n, float g) {
  // torch norm backward: self * (grad / norm), 0 where norm == 0.
  float s = (n == 0.0f) ? 0.0f : div_t<S::kFast>(g, n);
  return mk3(w.x * s, w.y * s, w.z * s);
}

template <class S, class PT>
RM_DEV void bwd_op(S& s, const PT& P, int op, int off, int a0, int a1) {
  const int A = s.acc0;
  switch (op) {
    case RM_OP_SPHERE: {
      float n = norm3_t<S::kFast>(s.p);
      s.gp = s.gp + safe_unit_scaled<S>(s.p, n, s.g);
      padd(s, A + off, -s.g);
    } break;
    case RM_OP_BOX: {
      float qx = fabsf(s.p.x) - P[off], qy = fabsf(s.p.y) - P[off + 1], qz = fabsf(s.p.z) - P[off + 2];
      float m = t_max(t_max(qx, qy), qz);
      V3 r = mk3(t_relu_keep(qx), t_relu_keep(qy), t_relu_keep(qz));
      float nr = norm3_t<S::kFast>(r);
      V3 gr = safe_unit_scaled<S>(r, nr, s.g);
      // where(q > 0): gradient only where q > 0 ; max(dim): first index attaining the max
      float gm = (m < 0.0f) ? s.g : 0.0f;
      int arg = (qx == m) ? 0 : ((qy == m) ? 1 : 2);
      float gqx = ((qx > 0.0f) ? gr.x : 0.0f) + ((arg == 0) ? gm : 0.0f);
      float gqy = ((qy > 0.0f) ? gr.y : 0.0f) + ((arg == 1) ? gm : 0.0f);
      float gqz = ((qz > 0.0f) ? gr.z : 0.0f) + ((arg == 2) ? gm : 0.0f);
      s.gp = s.gp + mk3(gqx * sgn0(s.p.x), gqy * sgn0(s.p.y), gqz * sgn0(s.p.z));
      padd(s, A + off, -gqx); padd(s, A + off + 1, -gqy); padd(s, A + off + 2, -gqz);
    } break;
    case RM_OP_PLANE:
      s.gp.x += s.g;
      break;
    case RM_OP_LINE: {
      V3 ab = P.v3(a0), abs_ = P.v3(a0 + 3);
      V3 ap = s.p - P.v3(off);
      float h0 = dot_seq(ap, abs_);
      float h = t_clamp(h0, 0.0f, 1.0f);
      V3 w = mk3(h * ab.x - ap.x, h * ab.y - ap.y, h * ab.z - ap.z);
      float nw = norm3_t<S::kFast>(w);
      V3 gw = safe_unit_scaled<S>(w, nw, s.g);
      float gh = (gw.x * ab.x + gw.y * ab.y) + gw.z * ab.z;
      float gh0 = (h0 >= 0.0f && h0 <= 1.0f) ? gh : 0.0f;  // clamp passes grad on the closed interval
      V3 gap = mk3(gh0 * abs_.x - gw.x, gh0 * abs_.y - gw.y, gh0 * abs_.z - gw.z);
      s.gp = s.gp + gap;
      padd(s, A + off, -gap.x); padd(s, A + off + 1, -gap.y); padd(s, A + off + 2, -gap.z);  // start
      padd(s, A + off + 6, -s.g);                                                               // radius
      padd(s, A + a0, h * gw.x); padd(s, A + a0 + 1, h * gw.y); padd(s, A + a0 + 2, h * gw.z);           // dAB
      padd(s, A + a0 + 3, gh0 * ap.x); padd(s, A + a0 + 4, gh0 * ap.y); padd(s, A + a0 + 5, gh0 * ap.z);  // d(AB/|AB|^2)
    } break;
    case RM_OP_DISK: {
      float a = norm2_t<S::kFast>(s.p.y, s.p.z);
      float rd = a - P[off];
      float c = t_relu_keep(rd);
      float d = norm2_t<S::kFast>(s.p.x, c);
      float sc = (d == 0.0f) ? 0.0f : div_t<S::kFast>(s.g, d);
      float gc = c * sc;
      float grd = (rd > 0.0f) ? gc : 0.0f;
      float sa = (a == 0.0f) ? 0.0f : div_t<S::kFast>(grd, a);
      s.gp = s.gp + mk3(s.p.x * sc, s.p.y * sa, s.p.z * sa);
      padd(s, A + off, -grd);
    } break;
    case RM_OP_TORUS: {
      float a = norm2_t<S::kFast>(s.p.x, s.p.z);
      float ring = a - P[off];
      float d0 = norm2_t<S::kFast>(ring, s.p.y);
      float sc = (d0 == 0.0f) ? 0.0f : div_t<S::kFast>(s.g, d0);
      float gring = ring * sc;
      float sa = (a == 0.0f) ? 0.0f : div_t<S::kFast>(gring, a);
      s.gp = s.gp + mk3(s.p.x * sa, s.p.y * sc, s.p.z * sa);
      padd(s, A + off, -gring);
      padd(s, A + off + 1, -s.g);
    } break;
    case RM_OP_AFFINE_POP: {  // reverse order: enter the child frame
      s.st->st(s.sp, s.p.x); s.st->st(s.sp + 1, s.p.y); s.st->st(s.sp + 2, s.p.z);
      s.st->st(s.sp + 3, s.gp.x); s.st->st(s.sp + 4, s.gp.y); s.st->st(s.sp + 5, s.gp.z);
      s.sp += 6;
      s.p = qrot(s.p - P.v3(off), P[off + 3], neg(P.v3(off + 4)));
      s.gp = mk3(0.0f, 0.0f, 0.0f);
    } break;
    case RM_OP_AFFINE_PUSH: {  // reverse order: leave the child frame, pull gp back
      V3 gl = s.gp;
      s.sp -= 6;
      V3 po = mk3(s.st->ld(s.sp), s.st->ld(s.sp + 1), s.st->ld(s.sp + 2));
      V3 gpo = mk3(s.st->ld(s.sp + 3), s.st->ld(s.sp + 4), s.st->ld(s.sp + 5));
      float w = P[off + 3];
      V3 u = neg(P.v3(off + 4));
      V3 v = po - P.v3(off);
      V3 t = 2.0f * cross(u, v);
      // local = v + w t + u x t
      V3 ugl = cross(u, gl);
      V3 gv = (gl + 2.0f * cross(u, ugl)) - (2.0f * w) * ugl;   // J_v^T gl
      float gw = (gl.x * t.x + gl.y * t.y) + gl.z * t.z;
      V3 gt = w * gl + cross(gl, u);
      V3 gu = cross(t, gl) + 2.0f * cross(v, gt);
      padd(s, A + off, -gv.x); padd(s, A + off + 1, -gv.y); padd(s, A + off + 2, -gv.z);
      padd(s, A + off + 3, gw);
      padd(s, A + off + 4, -gu.x); padd(s, A + off + 5, -gu.y); padd(s, A + off + 6, -gu.z);
      s.p = po;
      s.gp = gpo + gv;
    } break;
    case RM_OP_UNION_END: {  // reverse: open the frame; find the winner (first index on ties)
      s.st->st(s.sp, s.gframe); s.st->st(s.sp + 1, s.fval); s.sp += 2;
      float m = __builtin_inff();
      for (int i = 0; i < a1; ++i) m = t_min(m, s.st->ld(s.tape0 + a0 + i));
      int win = a0 + a1 - 1;
      for (int i = a1 - 1; i >= 0; --i) {
        float di = s.st->ld(s.tape0 + a0 + i);
        if (di == m || (m != m && di != di)) win = a0 + i;
      }
      s.gframe = s.g;
      s.fval = (float)win;
    } break;
    case RM_OP_FOLD_MIN:
      s.g = ((float)a0 == s.fval) ? s.gframe : 0.0f;
      break;
    case RM_OP_UNION_BEGIN:
      s.sp -= 2; s.gframe = s.st->ld(s.sp); s.fval = s.st->ld(s.sp + 1);
      break;
    case RM_OP_SMOOTH_END: {
      s.st->st(s.sp, s.gframe); s.st->st(s.sp + 1, s.fval); s.sp += 2;
      float k = P[off];
      float L = s.st->ld(s.tape0 + a0 + a1);     // logsumexp value recorded by the forward pass
      s.gframe = s.g;
      s.fval = L;
      // out = L / (-k): d out / dk through the division
      padd(s, A + off, div_t<S::kFast>(s.g * L, k * k));
    } break;
    case RM_OP_FOLD_LSE: {
      float k = P[off];
      float di = s.st->ld(s.tape0 + a0);
      // softmax weight of this child, exp(x_i - L) (autograd of logsumexp).  A gradient factor, not a value:
      // the hardware exponential (v_exp_f32, ~2 ulp) instead of the 21-instruction fp64 path -- 1e-7 relative
      // on a quantity whose contract is 1e-4 -- keeps 4 to 32 exponentials per VJP off the critical path.
      float w = __builtin_amdgcn_exp2f((di * (-k) - s.fval) * 1.44269504088896340736f);
      s.g = s.gframe * w;
      padd(s, A + off, div_t<S::kFast>(s.gframe, k) * w * di);
    } break;
    case RM_OP_SMOOTH_BEGIN:
      s.sp -= 2; s.gframe = s.st->ld(s.sp); s.fval = s.st->ld(s.sp + 1);
      break;
    case RM_OP_ROUND:
      padd(s, A + off, -s.g);
      break;
    case RM_OP_ONION: {
      padd(s, A + off, -s.g);
      s.g = s.g * sgn0(s.st->ld(s.tape0 + a0));
    } break;
    default:
      break;
  }
}

// --------------------------------------------------------------------------
// program drivers
// --------------------------------------------------------------------------
struct Ins {
  int op, off, a0, a1;
};

// Interpreter; instruction words are wave-uniform (scalar loads when `code` points at global memory).
struct RuntimeProgram {
  static constexpr int kTracked = 0;     // the interpreter always runs the full cull test
  static constexpr bool kNeedsFullWave = true;   // may contain CULL_LSE (wave-wide DPP reductions: every lane active)
  const int4* code;
  int n;
  template <class S, class PT>
  RM_DEV void forward(S& s, const PT& P) const {
    for (int pc = 0; pc < n; ++pc) {
      int4 w = code[pc];
      const int op = uniform_i(w.x), off = uniform_i(w.y), a0 = uniform_i(w.z), a1 = uniform_i(w.w);
      if (op == RM_OP_CULL_MIN) {            // a1 = (instructions up to and including the child's FOLD) << 8 | slot
        bool cull = cull_min_test(s, P, a0);
        if (!cull && off != 0) {             // the child is a smooth union with a bound table (its SMOOTH_BEGIN follows)
          const int4 sb = code[pc + 1];
          cull = cull_union_children(s, P, uniform_i(sb.y), uniform_i(sb.z), (uniform_i(sb.w) >> 8) & 255, uniform_i(sb.w) & 255);
        }
        if (cull) {
          const int slot = a1 & 255;
          if (s.record) { s.st->st(s.tape0 + slot, __builtin_inff()); s.culled |= 1ull << slot; }
          pc += a1 >> 8;
        }
        continue;
      }
      if (op == RM_OP_CULL_LSE) {            // a0 = tape slot of the child, a1 = instructions up to and including its FOLD_LSE
        if ((s.culled >> a0) & 1ull) pc += a1;
        continue;
      }
      fwd_op(s, P, op, off, a0, a1);
    }
  }
  template <class S, class PT>
  RM_DEV void backward(S& s, const PT& P) const {
    for (int pc = n - 1; pc >= 0; --pc) {
      int4 w = code[pc];
      const int op = uniform_i(w.x), off = uniform_i(w.y), a0 = uniform_i(w.z), a1 = uniform_i(w.w);
      if (op == RM_OP_FOLD_MIN && a1 > 0 && ((s.culled >> a0) & 1ull)) {   // a1 = distance back to its CULL_MIN
        s.g = 0.0f;
        pc -= a1;                              // jump over the skipped child and its CULL_MIN
        continue;
      }
      if (op == RM_OP_FOLD_LSE && a1 > 0 && ((s.culled >> a0) & 1ull)) {   // softmax weight exactly 0: nothing flows into the child
        s.g = 0.0f;
        pc -= a1;
        continue;
      }
      bwd_op(s, P, op, off, a0, a1);
    }
  }
};

// Compile-time program: Code::code[] is constexpr, recursion unrolls it.
template <class Code>
struct StaticProgram {
  // index of the CULL_MIN at `pc` among the tracked sites (outermost frame only: inside an affine frame the
  // local point moves by |M dp|, which an un-normalised quaternion can stretch), or -1
  static constexpr int tracked_site(int pc) {
    int depth = 0, k = 0;
    for (int i = 0; i < pc; ++i) {
      if (Code::code[i].op == RM_OP_AFFINE_PUSH) ++depth;
      if (Code::code[i].op == RM_OP_AFFINE_POP) --depth;
      if (Code::code[i].op == RM_OP_CULL_MIN && depth == 0) ++k;
    }
    return (depth == 0 && k < kCullTracked) ? k : -1;
  }
  static constexpr int count_tracked() {
    int k = 0;
    for (int i = 0; i < Code::n; ++i)
      if (Code::code[i].op == RM_OP_CULL_MIN && tracked_site(i) >= 0) ++k;
    return k;
  }
  static constexpr int kTracked = count_tracked();
  static constexpr bool has_wave_reductions() {
    for (int i = 0; i < Code::n; ++i)
      if (Code::code[i].op == RM_OP_CULL_LSE || (Code::code[i].op == RM_OP_CULL_MIN && Code::code[i].off != 0)) return true;
    return false;
  }
  static constexpr bool kNeedsFullWave = has_wave_reductions();   // the DPP reductions of the table tests need every lane of the wave active
  // does the smooth union that ends at `pc` carry a bound table (exact culling of its children, RM_OP_CULL_LSE)?
  static constexpr bool smooth_culled(int pc) {
    int depth = 0;
    for (int i = pc; i >= 0; --i) {
      if (Code::code[i].op == RM_OP_SMOOTH_END) ++depth;
      if (Code::code[i].op == RM_OP_SMOOTH_BEGIN && --depth == 0) return Code::code[i].a0 != 0 && (Code::code[i].a1 >> 16) != 0;
    }
    return false;
  }
  // executes instructions [PC, END)
  template <int PC, int END, class S, class PT>
  RM_DEV void fwd_range(S& s, const PT& P) const {
    if constexpr (PC < END) {
      constexpr Ins i = Code::code[PC];
      if constexpr (i.op == RM_OP_CULL_MIN) {
        constexpr int skip = i.a1 >> 8, slot = i.a1 & 255;
        constexpr int T = tracked_site(PC);
        bool cull;
        if constexpr (T >= 0) cull = cull_min_tracked<T>(s, P, i.a0); else cull = cull_min_test(s, P, i.a0);
        if constexpr (i.off != 0) {           // the child is a smooth union with a bound table (its SMOOTH_BEGIN follows)
          constexpr Ins sb = Code::code[PC + 1];
          if (!cull) cull = cull_union_children(s, P, sb.off, sb.a0, (sb.a1 >> 8) & 255, sb.a1 & 255);
        }
        if (cull) {
          if (s.record) { s.st->st(s.tape0 + slot, __builtin_inff()); s.culled |= 1ull << slot; }
        } else {
          fwd_range<PC + 1, PC + 1 + skip>(s, P);
        }
        fwd_range<PC + 1 + skip, END>(s, P);
      } else if constexpr (i.op == RM_OP_CULL_LSE) {
        if (!((s.culled >> i.a0) & 1ull)) fwd_range<PC + 1, PC + 1 + i.a1>(s, P);
        fwd_range<PC + 1 + i.a1, END>(s, P);
      } else if constexpr (i.op == RM_OP_SMOOTH_END) {
        smooth_end_static<i.a0, i.a1, smooth_culled(PC)>(s, P, i.off);
        fwd_range<PC + 1, END>(s, P);
      } else {
        fwd_op(s, P, i.op, i.off, i.a0, i.a1);
        fwd_range<PC + 1, END>(s, P);
      }
    }
  }
  // executes instructions (BEGIN, PC] in reverse order
  template <int PC, int BEGIN, class S, class PT>
  RM_DEV void bwd_range(S& s, const PT& P) const {
    if constexpr (PC > BEGIN) {
      constexpr Ins i = Code::code[PC];
      if constexpr ((i.op == RM_OP_FOLD_MIN || i.op == RM_OP_FOLD_LSE) && i.a1 > 0) {
        if ((s.culled >> i.a0) & 1ull) {
          s.g = 0.0f;
        } else {
          bwd_op(s, P, i.op, i.off, i.a0, i.a1);
          bwd_range<PC - 1, PC - i.a1>(s, P);     // the child; PC - a1 is its CULL_MIN / CULL_LSE
        }
        bwd_range<PC - i.a1 - 1, BEGIN>(s, P);
      } else {
        if constexpr (i.op != RM_OP_CULL_MIN && i.op != RM_OP_CULL_LSE) bwd_op(s, P, i.op, i.off, i.a0, i.a1);
        bwd_range<PC - 1, BEGIN>(s, P);
      }
    }
  }
  template <class S, class PT>
  RM_DEV void forward(S& s, const PT& P) const { fwd_range<0, Code::n>(s, P); }
  template <class S, class PT>
  RM_DEV void backward(S& s, const PT& P) const { bwd_range<Code::n - 1, -1>(s, P); }
};

// Scene evaluation context shared by all kernels of one block.
template <class Prog, class Store, class PT>
struct Scene {
  Prog prog;
  PT P;
  Store* st;
  int tape0;   // store index of tape slot 0 (= stack_floats)
  int acc0;    // store index of gradient accumulator 0 (= stack_floats + n_slots)
  const float* lds = nullptr;   // staged parameter block (raw + derived) in LDS: per-lane table reads of lse_cull_mask

  // bounds of the tracked cull sites at the point of the previous evaluation (per lane; NaN = unknown)
  mutable float cull_lo[kCullTracked] = {}, cull_hi[kCullTracked] = {};   // meaningful after the first evaluation (move = NaN)

  // f(p) where p is at most `move` away from the point of this context's previous evaluation (NaN: no
  // such knowledge).  Identical value; the knowledge only decides which cull tests can be skipped.
  RM_DEV float eval_near(V3 p, float move, bool record = false) const {
    Fwd<Store> s;
    s.p = p; s.d = 0.0f; s.acc = __builtin_inff(); s.sp = 0; s.tape0 = tape0; s.st = st; s.record = record;
    s.culled = 0ull; s.lds = lds;
#pragma unroll
    for (int k = 0; k < Prog::kTracked; ++k) {
      // move = NaN (nothing known: first evaluation of a tile, every 16th step) turns both bounds into NaN
      s.cull_lo[k] = cull_lo[k] - move;
      s.cull_hi[k] = cull_hi[k] + move;
    }
    prog.forward(s, P);
#pragma unroll
    for (int k = 0; k < Prog::kTracked; ++k) { cull_lo[k] = s.cull_lo[k]; cull_hi[k] = s.cull_hi[k]; }
    return s.d;
  }
  RM_DEV float eval(V3 p, bool record = false) const { return eval_near(p, __builtin_nanf(""), record); }
  // VJP at point p with upstream g: returns dL/dp, adds parameter grads into the accumulators.
  // `value` (optional) receives f(p) from the recording forward pass.
  RM_DEV V3 vjp(V3 p, float g, float* value = nullptr) const {
    Fwd<Store, (RM_FAST_VJP != 0)> f;
    f.p = p; f.d = 0.0f; f.acc = __builtin_inff(); f.sp = 0; f.tape0 = tape0; f.st = st; f.record = true;
    f.culled = 0ull; f.lds = lds;
#pragma unroll
    for (int k = 0; k < Prog::kTracked; ++k) { f.cull_lo[k] = __builtin_nanf(""); f.cull_hi[k] = __builtin_nanf(""); }
    prog.forward(f, P);
    if (value) *value = f.d;
    Bwd<Store> b;
    b.p = p; b.gp = mk3(0.0f, 0.0f, 0.0f); b.g = g; b.gframe = 0.0f; b.fval = 0.0f;
    b.sp = 0; b.tape0 = tape0; b.acc0 = acc0; b.st = st; b.culled = f.culled;
    prog.backward(b, P);
    return b.gp;
  }
  // dL/dp only (upstream g), no parameter gradients: the accumulator arithmetic is compiled out
  RM_DEV V3 vjp_point(V3 p, float g, float* value = nullptr) const {
    Fwd<Store, (RM_FAST_VJP != 0)> f;
    f.p = p; f.d = 0.0f; f.acc = __builtin_inff(); f.sp = 0; f.tape0 = tape0; f.st = st; f.record = true;
    f.culled = 0ull; f.lds = lds;
#pragma unroll
    for (int k = 0; k < Prog::kTracked; ++k) { f.cull_lo[k] = __builtin_nanf(""); f.cull_hi[k] = __builtin_nanf(""); }
    prog.forward(f, P);
    if (value) *value = f.d;
    last_culled = f.culled;
    Bwd<Store, false> b;
    b.p = p; b.gp = mk3(0.0f, 0.0f, 0.0f); b.g = g; b.gframe = 0.0f; b.fval = 0.0f;
    b.sp = 0; b.tape0 = tape0; b.acc0 = acc0; b.st = st; b.culled = f.culled;
    prog.backward(b, P);
    return b.gp;
  }
  // The full VJP (parameter gradients included) at the point of the LAST vjp_point / vjp of this context, without its
  // forward half: the reverse pass only reads the tape slots (fold inputs, onion inputs, the logsumexp value) and the
  // skip bits, and a reverse pass leaves them as they were (its own saves go to the stack region).  No scene
  // evaluation may lie in between.  Used by the converged-tail loop of march_reverse: one point-gradient pass at the
  // anchor, the recursion, then the parameter pass with the summed upstream.
  mutable unsigned long long last_culled = 0ull;
  RM_DEV V3 vjp_replay(V3 p, float g) const {
    Bwd<Store> b;
    b.p = p; b.gp = mk3(0.0f, 0.0f, 0.0f); b.g = g; b.gframe = 0.0f; b.fval = 0.0f;
    b.sp = 0; b.tape0 = tape0; b.acc0 = acc0; b.st = st; b.culled = last_culled;
    prog.backward(b, P);
    return b.gp;
  }
};

// --------------------------------------------------------------------------
// LDS staging of the scene block: raw parameters, then derived constants.
// Derived constants (capsule AB and AB/|AB|^2, primitives.py:52-54) are
// computed once per block instead of once per evaluation.
// --------------------------------------------------------------------------
RM_DEV float ld_t(const void* base, int64_t i, int dt);
// Bounding sphere (centre, radius) of the surface of the subtree encoded by instructions [begin, end),
// in the frame in which that subtree is evaluated: subtree(p) >= |p - c| - R for every p.  R = +inf
// when no finite bound is known (plane, strongly non-unit affine quaternion, non-positive blend_k, odd
// parameters).  Runs once per block on one thread, from the parameters already staged in LDS.
struct BoundFrame {
  float cx, cy, cz, R;   // union frames: enclosing sphere of the children folded so far
  float slope;           // union frames: smallest slope among them
  int n;                 // children folded; -1 for an affine frame
  int off;               // affine frame: parameter offset
};

constexpr int kBoundDepth = 12;

// Bound of the subtree [begin, end):  subtree(p) >= slope |p - c| - R  for every p, with 0.5 < slope <= 1.
//   out = {cx, cy, cz, R, slope};  R = +inf when no finite bound is known.
// slope < 1 comes from affine nodes whose quaternion is not unit: the reference applies
// V + w t + qv x t, t = 2 qv x V (quaternion.py:55-72) without normalising, which for q = |q| u is the
// normal matrix M = (1 - s) I + s R_u, s = |q|^2, whose smallest singular value is min(1, 2 s - 1).
// Also an UPPER bound around the same centre,  subtree(p) <= uslope |p - c| + Ru  (out[5] = Ru, out[6] = uslope; Ru = +inf
// when none is known): every primitive here is an exact distance (or, inside, minus one), so |d(p)| is the distance
// to the surface, which lies inside the sphere (c, R): d(p) <= |p - c| + R.  Rounding subtracts at most max(-r, 0),
// the onion |d| - r satisfies |d| <= uslope |p - c| + max(Ru, R); an affine frame stretches distances by at most the
// largest singular value max(1, 2 s - 1) of M.  Unions and smooth unions give none (min / -lse of the children's bounds
// would need one sphere per child); it is only used for the nearest-child estimate of lse_cull_mask.
template <class GetIns>
RM_DEV void subtree_bound(GetIns ins, const float* P, int begin, int end, float* out /*[7]*/, BoundFrame* st) {
  constexpr int kDepth = kBoundDepth;
  int sp = 0;
  float cx = 0.0f, cy = 0.0f, cz = 0.0f, R = __builtin_inff(), slope = 1.0f;
  float Ru = __builtin_inff(), uslope = 1.0f;
  const float inf = __builtin_inff();
  bool overflow = false;
  for (int pc = begin; pc < end && !overflow; ++pc) {
    const int4 w = ins(pc);
    const int op = w.x, off = w.y;
    switch (op) {
      case RM_OP_SPHERE: cx = cy = cz = 0.0f; slope = uslope = 1.0f; R = Ru = (P[off] >= 0.0f) ? P[off] : inf; break;
      case RM_OP_BOX: {
        float hx = P[off], hy = P[off + 1], hz = P[off + 2];
        cx = cy = cz = 0.0f; slope = uslope = 1.0f;
        R = Ru = (hx >= 0.0f && hy >= 0.0f && hz >= 0.0f) ? sqrtf(hx * hx + hy * hy + hz * hz) : inf;
      } break;
      case RM_OP_PLANE: slope = uslope = 1.0f; R = Ru = inf; break;
      case RM_OP_LINE: {   // capsule: sphere around the midpoint of AB
        const float* a = P + off;
        cx = 0.5f * (a[0] + a[3]); cy = 0.5f * (a[1] + a[4]); cz = 0.5f * (a[2] + a[5]);
        float hx = 0.5f * (a[3] - a[0]), hy = 0.5f * (a[4] - a[1]), hz = 0.5f * (a[5] - a[2]);
        slope = uslope = 1.0f;
        R = Ru = (a[6] >= 0.0f) ? sqrtf(hx * hx + hy * hy + hz * hz) * 1.00001f + a[6] : inf;
      } break;
      case RM_OP_DISK: cx = cy = cz = 0.0f; slope = uslope = 1.0f; R = Ru = (P[off] >= 0.0f) ? P[off] : inf; break;
      case RM_OP_TORUS:
        cx = cy = cz = 0.0f; slope = uslope = 1.0f;
        R = Ru = (P[off] >= 0.0f && P[off + 1] >= 0.0f) ? P[off] + P[off + 1] : inf;
        break;
      case RM_OP_ROUND:                                                         // d - r <= ub + max(-r, 0)
        Ru += fmaxf(-P[off], 0.0f);
        R += fmaxf(P[off], 0.0f);                                               // d - r >= d - max(r, 0)
        break;
      case RM_OP_ONION:                                                         // |d| <= uslope |p - c| + max(Ru, R)  (d >= -R)
        Ru = fmaxf(Ru, R) + fmaxf(-P[off], 0.0f);
        R += fmaxf(P[off], 0.0f);                                               // |d| - r >= d - max(r, 0)
        break;
      case RM_OP_AFFINE_PUSH:
        if (sp >= kDepth) { overflow = true; break; }
        st[sp].n = -1; st[sp].off = off; ++sp;
        break;
      case RM_OP_AFFINE_POP: {   // child frame -> parent frame: y = M (p - t), M = qrot(., conj(q))
        --sp;
        const float* a = P + st[sp].off;
        float w4 = a[3];
        V3 qv = mk3(a[4], a[5], a[6]);
        float s2 = ((w4 * w4 + qv.x * qv.x) + qv.y * qv.y) + qv.z * qv.z;
        float sigma = fminf(1.0f, 2.0f * s2 - 1.0f) - 1e-5f;           // smallest singular value of M, rounded down
        if (!(sigma > 0.5f) || !(s2 < 4.0f)) { R = Ru = inf; break; }
        // child(y) >= slope |y - c| - R.  Take c' = M'(c) (M' = qrot(., q); exactly M^-1 only for unit q) and
        // measure the miss e = |M c' - c|:  |y - c| >= |M (p - t - c')| - e >= sigma |p - (t + c')| - e.
        V3 c0 = mk3(cx, cy, cz);
        V3 c1 = qrot(c0, w4, qv);
        V3 back = qrot(c1, w4, neg(qv)) - c0;
        float e = sqrtf((back.x * back.x + back.y * back.y) + back.z * back.z);
        cx = c1.x + a[0]; cy = c1.y + a[1]; cz = c1.z + a[2];
        R = (R + e) * 1.0001f + 1e-4f * (fabsf(cx) + fabsf(cy) + fabsf(cz));
        slope = slope * sigma;
        // upper bound: |M (p - t) - c| <= smax |p - (t + c')| + e,  smax = max(1, 2 s - 1) rounded up
        Ru = (Ru + uslope * e) * 1.0001f + 1e-4f * (fabsf(cx) + fabsf(cy) + fabsf(cz));
        uslope = uslope * (fmaxf(1.0f, 2.0f * s2 - 1.0f) + 1e-5f);
      } break;
      case RM_OP_UNION_BEGIN: case RM_OP_SMOOTH_BEGIN:
        if (sp >= kDepth) { overflow = true; break; }
        st[sp].n = 0; st[sp].R = 0.0f; st[sp].slope = 1.0f; st[sp].cx = st[sp].cy = st[sp].cz = 0.0f; ++sp;
        break;
      case RM_OP_FOLD_MIN: case RM_OP_FOLD_LSE: {
        // child_i(p) >= slope_i |p - c_i| - R_i >= slope (|p - c| - |c - c_i|) - R_i  with slope = min slope_i <= 1,
        // so any sphere (c, R) that encloses all the spheres (c_i, R_i) bounds the min: smallest sphere around two.
        BoundFrame& f = st[sp - 1];
        if (f.n == 0) { f.cx = cx; f.cy = cy; f.cz = cz; f.R = R; f.slope = slope; }
        else {
          float dx = cx - f.cx, dy = cy - f.cy, dz = cz - f.cz;
          float d = sqrtf(dx * dx + dy * dy + dz * dz) * 1.00001f;
          if (!(f.R < inf) || !(R < inf)) { f.R = inf; }
          else if (d + R <= f.R) { /* already inside */ }
          else if (d + f.R <= R) { f.cx = cx; f.cy = cy; f.cz = cz; f.R = R; }
          else {
            float Rn = 0.5f * ((d + f.R) + R);
            float tt = (Rn - f.R) / d;                     // d > 0 here
            f.cx += tt * dx; f.cy += tt * dy; f.cz += tt * dz;
            f.R = Rn * 1.00001f + 1e-6f * (fabsf(f.cx) + fabsf(f.cy) + fabsf(f.cz));
          }
          f.slope = fminf(f.slope, slope);
        }
        f.n += 1;
      } break;
      case RM_OP_UNION_END:
        --sp; cx = st[sp].cx; cy = st[sp].cy; cz = st[sp].cz; R = st[sp].R; slope = st[sp].slope;
        Ru = inf; uslope = 1.0f;
        break;
      case RM_OP_SMOOTH_END: {   // -lse(-k d)/k >= min d - log(n)/k  for k > 0
        --sp; cx = st[sp].cx; cy = st[sp].cy; cz = st[sp].cz; R = st[sp].R; slope = st[sp].slope;
        Ru = inf; uslope = 1.0f;
        float k = P[off];
        R = (k > 0.0f) ? R + logf((float)st[sp].n) / k : inf;
      } break;
      default: break;            // nested CULL_MIN / CULL_LSE: no effect on the bound
    }
  }
  if (overflow || !(R == R) || !(slope > 0.5f)) R = inf;
  if (overflow || !(Ru == Ru) || !(uslope >= 1.0f) || !(uslope < 8.0f)) Ru = inf;
  out[0] = cx; out[1] = cy; out[2] = cz; out[3] = R; out[4] = slope; out[5] = Ru; out[6] = uslope;
}

// Derived constants of one instruction (capsule AB and AB/|AB|^2, primitives.py:52-54; bounding sphere
// of a cullable union child), computed from the staged parameters.
template <class GetIns>
RM_DEV void derive_line_constants(GetIns ins, int pc, float* s_params) {
  const int4 w = ins(pc);
  if (w.x == RM_OP_LINE) {
    const float* a = s_params + w.y;
    float abx = a[3] - a[0], aby = a[4] - a[1], abz = a[5] - a[2];
    float len2 = (abx * abx + aby * aby) + abz * abz;  // AB.pow(2).sum(-1)
    float* dst = s_params + w.z;
    dst[0] = abx; dst[1] = aby; dst[2] = abz;
    dst[3] = abx / len2; dst[4] = aby / len2; dst[5] = abz / len2;
  }
}

// all derived constants of a program; call with every thread of the block (contains barriers)
template <class GetIns>
RM_DEV void derive_constants(GetIns ins, int n_instr, float* s_params) {
  // Bounding spheres: a walk over the subtree per cull site, a chain of dependent LDS reads and scalar-like flops on one
  // lane (~1 us per instruction visited: 9 us for closed scene 1, 120 us for the 32-primitive scene with its bound
  // table).  The sites are dealt round robin to the first lanes of the block's waves, each with its own stack.
  constexpr int kWalkers = 4;
  __shared__ BoundFrame s_bound_stacks[kWalkers][kBoundDepth];
  for (int i = threadIdx.x; i < n_instr; i += blockDim.x) derive_line_constants(ins, i, s_params);
  const int walker = threadIdx.x >> 6;
  const int n_walkers = ((int)(blockDim.x >> 6) < kWalkers) ? (int)(blockDim.x >> 6) : kWalkers;
  if ((threadIdx.x & 63) == 0 && walker < n_walkers) {
    BoundFrame* s_bound_stack = s_bound_stacks[walker];
    int site = 0;
    auto mine = [&]() { return (site++ % n_walkers) == walker; };
    for (int pc = 0; pc < n_instr; ++pc) {
      const int4 w = ins(pc);
      if (w.x == RM_OP_SMOOTH_BEGIN && w.z > 0) {
        // bound table of a smooth union, one entry per child in slot order: {cx, cy, cz, slope_lb, K_lb, slope_ub, K_ub, 0},
        // the lower bound with the margins of CULL_MIN, the upper one with the same ones on the other side.  Children are
        // the instruction ranges between the FOLD_LSEs of this union's own level (a CULL_LSE in front is skipped).
        int q = pc + 1;
        for (int j = 0; j < (w.w & 255); ++j) {
          if (ins(q).x == RM_OP_CULL_LSE) ++q;
          int end = q, depth = 0;
          for (;; ++end) {
            const int op = ins(end).x;
            if (op == RM_OP_UNION_BEGIN || op == RM_OP_SMOOTH_BEGIN) ++depth;
            else if (op == RM_OP_UNION_END || op == RM_OP_SMOOTH_END) --depth;
            else if (op == RM_OP_FOLD_LSE && depth == 0) break;
          }
          if (!mine()) { q = end + 1; continue; }
          float b[7];
          subtree_bound(ins, s_params, q, end, b, s_bound_stack);
          float* out = s_params + w.z + 8 * j;
          const float c1 = (fabsf(b[0]) + fabsf(b[1])) + fabsf(b[2]);
          const float Kl = ((b[3] * 1.0001f + 1e-4f) + 1e-5f * c1) * 1.000001f;
          const float Ku = ((b[5] * 1.0001f + 1e-4f) + 1e-5f * c1) * 1.000001f;
          out[0] = b[0]; out[1] = b[1]; out[2] = b[2];
          out[3] = b[4] - 2e-4f;
          out[4] = Kl < __builtin_inff() ? Kl : __builtin_nanf("");
          out[5] = b[6] + 2e-4f;
          out[6] = Ku < __builtin_inff() ? Ku : __builtin_inff();
          out[7] = 0.0f;
          q = end + 1;
        }
      }
      if (w.x == RM_OP_CULL_MIN && mine()) {
        float b7[7];
        float* out = s_params + w.z;
        subtree_bound(ins, s_params, pc + 1, pc + (w.w >> 8), b7, s_bound_stack);  // child without its FOLD
        for (int q = 0; q < 5; ++q) out[q] = b7[q];
        // cull_min_test:  (slope - 2e-4) dist - K >= acc  with  K = 1.0001 R + 1e-4 : the bound with a margin of
        // 1e-4 (1 + 2 dist + R), three orders of magnitude above the fp32 rounding of the child's own value
        // (+ 1e-5 |c|_1: rounding drift of p over the <= 16 steps a tracked bound is carried, eval_near)
        float K = ((out[3] * 1.0001f + 1e-4f) + 1e-5f * ((fabsf(out[0]) + fabsf(out[1])) + fabsf(out[2]))) * 1.000001f;
        out[3] = K < __builtin_inff() ? K : __builtin_nanf("");
        out[4] = out[4] - 2e-4f;
      }
    }
  }
  __syncthreads();
}

// LDS staging of the scene block: raw parameters, then derived constants.
// raw parameter block -> LDS: from the packed fp32 array, or gathered from the parameter tensors themselves
RM_DEV void stage_params(const RmScene& sc, float* s_params, int n_params) {
  if (sc.param_refs) {
    for (int i = threadIdx.x; i < n_params; i += blockDim.x) {
      const RmParamRef r = sc.param_refs[i];
      s_params[i] = ld_t(r.base, r.elem, r.dtype);
    }
  } else {
    for (int i = threadIdx.x; i < n_params; i += blockDim.x) s_params[i] = sc.params[i];
  }
}

// RmScene::block: the finished block of an earlier launch instead of gather + derive_constants (rm_abi.h)
RM_DEV void load_scene_block(const RmScene& sc, float* s_params) {
  for (int i = threadIdx.x; i < sc.n_params + sc.n_derived; i += blockDim.x) s_params[i] = sc.block[i];
}
// RmScene::block_out: block 0 leaves its staged block for later launches (call behind derive_constants' last barrier)
RM_DEV void store_scene_block(const RmScene& sc, const float* s_params) {
  if (sc.block_out && blockIdx.x == 0)
    for (int i = threadIdx.x; i < sc.n_params + sc.n_derived; i += blockDim.x) sc.block_out[i] = s_params[i];
}

// RmScene::block_cache (rm_abi.h): the derived constants of an earlier launch, reused when the parameters they were derived
// from are, bit for bit, the ones just gathered.  Returns true when s_params holds the finished block.  Call with the raw
// parameters staged (no barrier needed before); ends behind a barrier.
RM_DEV bool try_scene_cache(const RmScene& sc, float* s_params) {
  if (!sc.block_cache || sc.n_params <= 0) return false;
  bool same = true;      // (every thread compares the parameters it staged itself)
  for (int i = threadIdx.x; i < sc.n_params; i += blockDim.x) {
    const float c = __hip_atomic_load(sc.block_cache + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    same = same && (__builtin_bit_cast(unsigned, c) == __builtin_bit_cast(unsigned, s_params[i]));
  }
  if (!__syncthreads_and(same)) return false;
  // No fence here (an agent-scope fence writes back and invalidates caches: 60 us per 1080p frame when every block did
  // it): these loads are issued after the vote, i.e. after the parameter loads have returned, and both go to the
  // device-coherent L2, where the derived constants were written before the parameters that vouch for them.
  for (int i = sc.n_params + threadIdx.x; i < sc.n_params + sc.n_derived; i += blockDim.x)
    s_params[i] = __hip_atomic_load(sc.block_cache + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  return true;
}
// ... and block 0 of a launch that had to derive them leaves them for the next one (behind derive_constants' last barrier)
RM_DEV void fill_scene_cache(const RmScene& sc, const float* s_params) {
  if (!sc.block_cache || sc.n_params <= 0 || blockIdx.x != 0) return;
  for (int i = sc.n_params + threadIdx.x; i < sc.n_params + sc.n_derived; i += blockDim.x)
    __hip_atomic_store(sc.block_cache + i, s_params[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __threadfence();
  __syncthreads();
  for (int i = threadIdx.x; i < sc.n_params; i += blockDim.x)
    __hip_atomic_store(sc.block_cache + i, s_params[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

RM_DEV void stage_scene(const RmScene& sc, float* s_params, int4* s_prog) {
  const int4* gprog = reinterpret_cast<const int4*>(sc.program);
  for (int i = threadIdx.x; i < sc.n_instr; i += blockDim.x) s_prog[i] = gprog[i];
  if (sc.block) {
    load_scene_block(sc, s_params);
    __syncthreads();
  } else {
    stage_params(sc, s_params, sc.n_params);
    if (!try_scene_cache(sc, s_params)) {
      __syncthreads();
      auto ins = [s_prog](int pc) { return s_prog[pc]; };
      derive_constants(ins, sc.n_instr, s_params);
      fill_scene_cache(sc, s_params);
    }
  }
  store_scene_block(sc, s_params);
}

// --------------------------------------------------------------------------
// camera, normals, shaders
// --------------------------------------------------------------------------
struct Pose {
  float w;
  V3 qv;
  V3 t;
};

RM_DEV V3 load3(const float* base, int64_t i) {
  const float* q = base + 3 * i;
  return V3{q[0], q[1], q[2]};
}
RM_DEV void store3(float* base, int64_t i, V3 v) {
  float* q = base + 3 * i;
  q[0] = v.x; q[1] = v.y; q[2] = v.z;
}

// typed I/O (RM_DTYPE_*): `dt` is a kernel argument, hence wave-uniform: a scalar branch per access
RM_DEV float ld_t(const void* base, int64_t i, int dt) {
  return (dt == RM_DTYPE_F16) ? (float)static_cast<const _Float16*>(base)[i] : static_cast<const float*>(base)[i];
}
RM_DEV void st_t(void* base, int64_t i, float v, int dt) {
  if (dt == RM_DTYPE_F16) static_cast<_Float16*>(base)[i] = (_Float16)v;          // round to nearest even, like .to(float16)
  else if (dt == RM_DTYPE_F64) static_cast<double*>(base)[i] = (double)v;
  else static_cast<float*>(base)[i] = v;
}
RM_DEV V3 load3_t(const void* base, int64_t i, int dt) {
  if (dt == RM_DTYPE_F16) {
    const _Float16* q = static_cast<const _Float16*>(base) + 3 * i;
    return V3{(float)q[0], (float)q[1], (float)q[2]};
  }
  return load3(static_cast<const float*>(base), i);
}
RM_DEV void store3_t(void* base, int64_t i, V3 v, int dt) {
  st_t(base, 3 * i, v.x, dt); st_t(base, 3 * i + 1, v.y, dt); st_t(base, 3 * i + 2, v.z, dt);
}

struct Tetra {
  V3 o[4];
  float inv[9];
  float lap_scale;
};

RM_DEV Tetra load_tetra(const RmTetra& t) {
  Tetra r;
  for (int k = 0; k < 4; ++k) r.o[k] = mk3(t.offsets[3 * k], t.offsets[3 * k + 1], t.offsets[3 * k + 2]);
  for (int k = 0; k < 9; ++k) r.inv[k] = t.inverse[k];
  r.lap_scale = t.lap_scale;
  return r;
}

// SDFNormals.forward (ray_marching.py:115-125).  `centre` = scene(p) is supplied by the caller.
template <class SceneT>
RM_DEV void normals_from_taps(const Tetra& T, float f0, float f1, float f2, float f3, float centre, V3& n, float& lap,
                              V3* u_out = nullptr) {
  float d1 = f1 - f0, d2 = f2 - f0, d3 = f3 - f0;
  V3 u = mk3((T.inv[0] * d1 + T.inv[1] * d2) + T.inv[2] * d3,
             (T.inv[3] * d1 + T.inv[4] * d2) + T.inv[5] * d3,
             (T.inv[6] * d1 + T.inv[7] * d2) + T.inv[8] * d3);
  if (u_out) *u_out = u;
  float nu = norm3(u);              // F.normalize(eps=0): u / |u|, 0/0 = NaN like the reference
#ifdef RM_FAST_MATH
  float inv = __builtin_amdgcn_rcpf(nu);
  n = mk3(u.x * inv, u.y * inv, u.z * inv);
#else
  n = mk3(u.x / nu, u.y / nu, u.z / nu);
#endif
  float mean = (((f0 + f1) + f2) + f3) / 4.0f;
  lap = (centre - mean) * T.lap_scale;
}

// The four tap evaluations run as ONE rolled loop (a single inlined copy of the scene evaluator
// instead of four: the interpreter body is large and the instruction cache is shared by two CUs).
template <class SceneT>
RM_DEV void eval_taps(const SceneT& sc, const Tetra& T, V3 p, float& f0, float& f1, float& f2, float& f3) {
  f0 = f1 = f2 = f3 = 0.0f;
  // offsets as scalars: selecting among array elements makes LLVM spill the array to scratch and index it
  const float ax = T.o[0].x, ay = T.o[0].y, az = T.o[0].z, bx = T.o[1].x, by = T.o[1].y, bz = T.o[1].z;
  const float cx = T.o[2].x, cy = T.o[2].y, cz = T.o[2].z, dx = T.o[3].x, dy = T.o[3].y, dz = T.o[3].z;
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const V3 ok = mk3((k == 0) ? ax : ((k == 1) ? bx : ((k == 2) ? cx : dx)),
                      (k == 0) ? ay : ((k == 1) ? by : ((k == 2) ? cy : dy)),
                      (k == 0) ? az : ((k == 1) ? bz : ((k == 2) ? cz : dz)));
    const float fk = sc.eval(p + ok);
    f0 = (k == 0) ? fk : f0; f1 = (k == 1) ? fk : f1; f2 = (k == 2) ? fk : f2; f3 = (k == 3) ? fk : f3;
  }
}

template <class SceneT>
RM_DEV void normals_forward(const SceneT& sc, const Tetra& T, V3 p, float centre, V3& n, float& lap, V3* u_out = nullptr) {
  float f0, f1, f2, f3;
  eval_taps(sc, T, p, f0, f1, f2, f3);
  normals_from_taps<SceneT>(T, f0, f1, f2, f3, centre, n, lap, u_out);
}

// monotone float <-> uint map for atomic min/max
RM_DEV uint32_t f2ord(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
RM_DEV float ord2f(uint32_t u) {
  u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  return __builtin_bit_cast(float, u);
}

}  // namespace rm

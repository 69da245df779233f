// rm_kernels.h -- gfx950 kernels over the scene machine of rm_device.h.
//
// Every kernel is a template over a configuration `Cfg` that says how the
// scene program is driven (RuntimeProgram interpreter + LDS store, or a
// StaticProgram + register store).  The generic library instantiates them with
// GenericCfg; per-scene specialisations (rm_spec.hip) instantiate the same
// kernels with a StaticCfg.
//
// Launch shape: one ray per lane, persistent blocks that stride over 'tiles' of
// blockDim rays; lanes past the end of the ray list keep executing on a clamped
// index (so wave-wide votes and scalar reads stay well defined) and only their
// stores are masked.
#pragma once

#include "rm_device.h"

#include <type_traits>

namespace rm {

// ---------------------------------------------------------------------------
// configurations
// ---------------------------------------------------------------------------
// Dynamic LDS layout of the generic path:
//   [ params + derived | program (int4) | per-thread store columns ]
struct GenericCfg {
  using Store = LdsStore;
  using Prog = RuntimeProgram;
  using SceneT = Scene<Prog, Store, LdsParams>;
  static constexpr bool kStatic = false;
  static constexpr bool kRowAcc = true;
  static RM_DEV int n_acc(const RmScene& sc) { return sc.n_params + sc.n_grad_derived; }
  // LDS floats of a block of `block` threads: parameter block, program, per thread one column (+1 of padding) of
  // evaluation stack and tape, and -- backward -- one row of gradient accumulators per wave (LdsStore::acc_row).  The
  // backward kernels fetch instructions through the scalar cache, so the staged program only serves derive_constants
  // and shares its space with the columns and rows (setup() returns behind derive_constants' last barrier).
  static __host__ __device__ size_t lds_floats(const RmScene& sc, int block, bool backward) {
    const size_t pb = (size_t)((sc.n_params + sc.n_derived + 3) & ~3), prog = 4 * (size_t)sc.n_instr;
    const size_t columns = ((size_t)sc.stack_floats + sc.n_slots) * (block + 1);
    if (!backward) return pb + prog + columns;
    const size_t work = columns + (size_t)(sc.n_params + sc.n_grad_derived) * (block >> 6);
    return pb + (prog > work ? prog : work);
  }

  // returns the scene context; `store` must outlive it
  static RM_DEV SceneT setup(const RmScene& sc, float* smem, Store& store, bool scalar_fetch = false) {
    int pb = (sc.n_params + sc.n_derived + 3) & ~3;
    float* s_params = smem;
    int4* s_prog = reinterpret_cast<int4*>(smem + pb);
    float* s_store = smem + pb + (scalar_fetch ? 0 : 4 * sc.n_instr);
    stage_scene(sc, s_params, s_prog);
    store.base = s_store + threadIdx.x;
    store.stride = blockDim.x + 1;
    store.acc0 = sc.stack_floats + sc.n_slots;
    store.acc_row = scalar_fetch      // (= a backward kernel)
        ? s_store + (size_t)(sc.stack_floats + sc.n_slots) * (blockDim.x + 1) + (threadIdx.x >> 6) * n_acc(sc) : nullptr;
    SceneT s;
    // Instruction fetch: LDS broadcast read + readfirstlane in the forward kernels; in the backward
    // kernels, where LDS is busy with the per-thread gradient accumulators, straight from the program
    // buffer through the scalar cache (kernel-argument pointer + wave-uniform pc -> s_load_dwordx4).
    // Measured on the interpreter: forward 1.66 ms (LDS) vs 1.82 (scalar); backward 2.44 vs 1.75.
    s.prog.code = scalar_fetch ? reinterpret_cast<const int4*>(sc.program) : s_prog;
    s.prog.n = sc.n_instr;
    s.P.p = s_params;
    s.lds = s_params;
    s.st = &store;
    s.tape0 = sc.stack_floats;
    s.acc0 = sc.stack_floats + sc.n_slots;
    return s;
  }
};

// Compile-time scene: Code supplies n, code[], n_params, n_derived, n_grad_derived, stack_floats, n_slots.
#ifndef RM_LDS_TAPE_MIN_SLOTS
#define RM_LDS_TAPE_MIN_SLOTS 17   // scenes with at least this many tape slots keep the tape in LDS columns (HybridStore)
#endif
#ifndef RM_STATIC_REG_ACC_MAX
#define RM_STATIC_REG_ACC_MAX 96
#endif
template <class Code, int kRegParamLimit = 64, bool kVgprParams = false>
struct StaticCfg {
  static constexpr int kAcc = Code::n_params + Code::n_grad_derived;     // gradient accumulators
  // up to RM_STATIC_REG_ACC_MAX of them ride in registers, one set per ray (closed scene 1: 46); more would spill, so
  // larger scenes keep one row per wave in LDS like the interpreter (RowAccStore)
  static constexpr bool kRowAcc = kAcc > RM_STATIC_REG_ACC_MAX;
  static constexpr int kStoreN = Code::stack_floats + Code::n_slots + (kRowAcc ? 0 : kAcc);
  static constexpr bool kLdsTape = Code::n_slots >= RM_LDS_TAPE_MIN_SLOTS;
  using BaseStore = std::conditional_t<kLdsTape, HybridStore<Code::stack_floats, Code::n_slots, kStoreN>, RegStore<kStoreN>>;
  using Store = std::conditional_t<kRowAcc, RowAccStore<BaseStore>, BaseStore>;
  using Prog = StaticProgram<Code>;
  // small parameter blocks ride in registers; big ones (config 5: 381 floats) stay in LDS
  static constexpr int kParamFloats = Code::n_params + Code::n_derived;
  static constexpr bool kRegParams = kParamFloats <= kRegParamLimit;
  using PT = std::conditional_t<kRegParams, RegParams<kParamFloats, kVgprParams>, LdsParams>;
  using SceneT = Scene<Prog, Store, PT>;
  static constexpr bool kStatic = true;
  static RM_DEV int n_acc(const RmScene&) { return kAcc; }

  // floats of the block's tape area (after the parameter block and, in backward kernels, the reduction scratch)
  static constexpr int kTapeFloats = kLdsTape ? Code::n_slots * kLdsTapeStride : 0;

  static RM_DEV SceneT setup(const RmScene& sc, float* smem, Store& store, bool backward = false) {
    // the parameter block is still staged through LDS (raw + derived), the program is not
    float* s_params = smem;
    constexpr int pb = (Code::n_params + Code::n_derived + 3) & ~3;
    if constexpr (kLdsTape) store.base = smem + pb + 4 + (backward ? (int)(blockDim.x >> 6) * kAcc : 0) + threadIdx.x;
    if constexpr (kRowAcc) {      // the rows are the block-reduction scratch of the register form (flush_accumulators)
      store.acc0 = Code::stack_floats + Code::n_slots;
      store.acc_row = backward ? smem + pb + (int)(threadIdx.x >> 6) * kAcc : nullptr;
    }
    if (sc.block) {
      load_scene_block(sc, s_params);
      __syncthreads();
    } else {
      stage_params(sc, s_params, Code::n_params);
      if (!try_scene_cache(sc, s_params)) {
        __syncthreads();
        auto ins = [](int pc) { const Ins& i = Code::code[pc]; return make_int4(i.op, i.off, i.a0, i.a1); };
        derive_constants(ins, Code::n, s_params);
        fill_scene_cache(sc, s_params);
      }
    }
    store_scene_block(sc, s_params);
    SceneT s;
    if constexpr (kRegParams) s.P.load(s_params); else s.P.p = s_params;
    s.lds = s_params;
    s.st = &store;
    s.tape0 = Code::stack_floats;
    s.acc0 = Code::stack_floats + Code::n_slots;
    return s;
  }
};

extern __shared__ __attribute__((aligned(16))) float rm_smem[];

// occupancy targets of the register-heavy backward kernels (A/B knobs; unset = the compiler's own choice)
#ifdef RM_BWD_WAVES_PER_EU
#define RM_BWD_OCC __attribute__((amdgpu_waves_per_eu(RM_BWD_WAVES_PER_EU, RM_BWD_WAVES_PER_EU)))
#else
#define RM_BWD_OCC
#endif
#ifdef RM_HARDB_WAVES_PER_EU
#define RM_HARDB_OCC __attribute__((amdgpu_waves_per_eu(RM_HARDB_WAVES_PER_EU, RM_HARDB_WAVES_PER_EU)))
#else
#define RM_HARDB_OCC
#endif

// ---------------------------------------------------------------------------
// gradient accumulators: zero at start, block-reduce into partials at the end
// ---------------------------------------------------------------------------
template <class Cfg>
RM_DEV void zero_accumulators(const typename Cfg::SceneT& sc, int n_acc) {
  if constexpr (Cfg::kRowAcc) {
    // this wave's row (LDS operations of one wave complete in order: no barrier before the first accumulation)
    for (int i = threadIdx.x & 63; i < n_acc; i += 64) sc.st->acc_row[i] = 0.0f;
  } else {
#pragma unroll
    for (int i = 0; i < Cfg::kAcc; ++i) sc.st->st(sc.acc0 + i, 0.0f);   // constant indices: stays in VGPRs
  }
}

// partials[blockIdx.x][n_acc]: deterministic (fixed lane order) sum over the block.
template <class Cfg>
RM_DEV void flush_accumulators(const typename Cfg::SceneT& sc, int n_acc, float* partials, float* smem_scratch) {
  if constexpr (Cfg::kRowAcc) {
    __syncthreads();
    const float* rows = sc.st->acc_row - (threadIdx.x >> 6) * n_acc;      // one row per wave, added in wave order
    for (int i = threadIdx.x; i < n_acc; i += blockDim.x) {
      float sum = 0.0f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) sum += rows[w * n_acc + i];
      partials[(int64_t)blockIdx.x * n_acc + i] = sum;
    }
  } else {
    // registers: butterfly over the wave, then waves through LDS scratch
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < Cfg::kAcc; ++i) {
      float v = sc.st->ld(sc.acc0 + i);
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (lane == 0) smem_scratch[wave * n_acc + i] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_acc; i += blockDim.x) {
      float sum = 0.0f;
      for (int w = 0; w < nw; ++w) sum += smem_scratch[w * n_acc + i];
      partials[(int64_t)blockIdx.x * n_acc + i] = sum;
    }
  }
}

// ---------------------------------------------------------------------------
// scene(query) forward / backward
// ---------------------------------------------------------------------------
template <class Cfg>
__global__ void __launch_bounds__(256) k_sdf_fwd(RmScene sc, const void* __restrict__ pts, void* __restrict__ dist, int64_t n, int dt) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(sc, rm_smem, store);
  int64_t ntiles = (n + blockDim.x - 1) / blockDim.x;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t i = tile * blockDim.x + threadIdx.x;
    bool live = i < n;
    int64_t ic = live ? i : n - 1;
    float d = scene.eval(load3_t(pts, ic, dt));
    if (live) st_t(dist, i, d, dt);
  }
}

template <class Cfg>
__global__ void __launch_bounds__(256) k_sdf_bwd(RmScene sc, const float* __restrict__ pts, const float* __restrict__ gd,
                          float* __restrict__ gpts, float* __restrict__ partials, int64_t n) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(sc, rm_smem, store, true);
  const int n_acc = Cfg::n_acc(sc);
  zero_accumulators<Cfg>(scene, n_acc);
  int64_t ntiles = (n + blockDim.x - 1) / blockDim.x;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t i = tile * blockDim.x + threadIdx.x;
    bool live = i < n;
    int64_t ic = live ? i : n - 1;
    float g = live ? gd[ic] : 0.0f;
    V3 gp = scene.vjp(load3(pts, ic), g);
    if (live && gpts) store3(gpts, i, gp);
  }
  flush_accumulators<Cfg>(scene, n_acc, partials, rm_smem + ((sc.n_params + sc.n_derived + 3) & ~3));
}

// ---------------------------------------------------------------------------
// march
// ---------------------------------------------------------------------------
RM_DEV bool same_bits(V3 a, V3 b) {
  return (__builtin_bit_cast(int, a.x) == __builtin_bit_cast(int, b.x)) &
         (__builtin_bit_cast(int, a.y) == __builtin_bit_cast(int, b.y)) &
         (__builtin_bit_cast(int, a.z) == __builtin_bit_cast(int, b.z));
}

// SDFMarcher.forward (ray_marching.py:78-84): p <- f(p)*v + p, `steps` times.
//
// Early-out (bit-exact).  The march map is a pure function of p (v is fixed per ray), so a
// repeated state proves a cycle: if p_{i+1} equals a remembered earlier iterate p_s bitwise,
// every later iterate is p_{s + ((j - s) mod lambda)} with lambda = i + 1 - s.  Each lane keeps
// one snapshot, refreshed at power-of-two steps (Brent's cycle finder), plus the immediate
// fixed-point test p_{i+1} == p_i (lambda = 1).  Near a surface fp32 iterates stop moving or
// hop between a few neighbouring values, so after convergence every ray is in such a cycle
// (measured: without the snapshot test 30% of wave tiles ran all 128 steps of config 2).
// The wave leaves when ALL 64 rays have a known period; each lane is then advanced by
// (remaining mod lambda) < lambda further steps so it lands on exactly the iterate the full
// loop would have produced.  With a trajectory being recorded (backward) the same exits are taken and the
// iterates of steps >= nexec are not stored: the reverse sweep uses p_final for them (exact for a fixed point,
// an ulp or two off inside a cycle between neighbouring floats -- a VJP argument, tolerance 1e-4).
#ifndef RM_PRIO_TILE
#define RM_PRIO_TILE 32           // march step from which a tile's wave runs at raised issue priority (0 = never)
#endif
#ifndef RM_PRIO_REGEN
#define RM_PRIO_REGEN 16          // the same for a pool holding a ray that has marched this long
#endif
#ifndef RM_EARLY_DENSE_STEPS
#define RM_EARLY_DENSE_STEPS 8    // look for cycles after every 2nd step up to here, after every 4th from then on
#endif                            // (>= 4: the snapshot refreshes at steps 2 and 4 happen inside those looks)

// p + f v, each component one multiplication and one addition (the reference's `distances * directions + positions`,
// no contraction).  -DRM_PK_UPDATE: x and y through the packed fp32 forms (v_pk_mul_f32 / v_pk_add_f32: the same IEEE
// results, 4 instructions instead of 6); measured 1080p scene 2 207 -> 204 us, pools unchanged: inside the noise, off
typedef float rm_v2f __attribute__((ext_vector_type(2)));
RM_DEV V3 step_point(V3 p, V3 v, float f) {
#ifdef RM_PK_UPDATE
  const rm_v2f vxy = {v.x, v.y}, pxy = {p.x, p.y}, ff = {f, f};
  const rm_v2f r = ff * vxy + pxy;
  return mk3(r.x, r.y, f * v.z + p.z);
#else
  return mk3(f * v.x + p.x, f * v.y + p.y, f * v.z + p.z);
#endif
}

// Where a recorded march keeps its iterates.  kSoa = false: [step][ray][3], stride = rays (rm_march_forward: any list
// of rays).  kSoa = true (the fused frame): [wave tile][step][component][lane], stride = the step count, idx = the
// ray's slot = tile * 64 + lane: each of the three stores of a wave is 256 contiguous bytes (per-pixel [ray][3] rows of an
// 8x8 tile are eight 96-byte pieces), and the whole trajectory of a tile is ONE contiguous block of steps x 768 bytes,
// so the forward's successive steps and the reverse sweep's walk stay inside a few pages instead of touching three new
// ones, megabytes apart, at every step.
template <bool kSoa>
RM_DEV void traj_store(float* traj, int64_t stride, int64_t idx, int i, V3 p) {
  if constexpr (kSoa) {
    float* q = traj + ((idx >> 6) * stride + i) * 192 + (idx & 63);
    q[0] = p.x; q[64] = p.y; q[128] = p.z;
  } else {
    store3(traj + 3 * (int64_t)i * stride, idx, p);
  }
}
template <bool kSoa>
RM_DEV V3 traj_load(const float* traj, int64_t stride, int64_t idx, int i) {
  if constexpr (kSoa) {
    const float* q = traj + ((idx >> 6) * stride + i) * 192 + (idx & 63);
    return mk3(q[0], q[64], q[128]);
  } else {
    return load3(traj + 3 * (int64_t)i * stride, idx);
  }
}

struct NoPark {
  static constexpr bool kEnabled = false;
  RM_DEV bool operator()(int, V3, bool) const { return false; }
};

#ifndef RM_PARK_MAX_LANES
#define RM_PARK_MAX_LANES 40      // park only a minority: a tile whose rays are (nearly) all still moving stays put
#endif
#ifndef RM_PARK_FIRST
#define RM_PARK_FIRST 2           // first check step, in strides (a ray parked early may have settled a few steps later)
#endif

// check steps of the parking lists: stride 16 for <= 128 steps, list k <-> step (k + 2) * stride
RM_DEV int park_stride(int steps) { return steps <= 128 ? 16 : ((steps + 127) / 128) * 16; }

// `park(step, p_next, unsettled)`: called wave-wide at a check step when 1..RM_PARK_MAX_LANES rays of the wave are
// not yet in a proven cycle; returns (per lane) whether the ray was put on a list -- it then belongs to
// k_render_parked and this wave neither waits for it nor stores its pixel (`parked` out).
template <class SceneT, class ParkF = NoPark, bool kSoa = false>
RM_DEV V3 march(const SceneT& scene, V3 p, V3 v, int steps, bool early, float* traj, int64_t traj_stride,
                int64_t ray, bool live, int& nexec, ParkF park = ParkF(), bool* parked = nullptr) {
  if (parked) *parked = false;
  const int pstride = park_stride(steps);
  V3 snap = p;          // remembered iterate p_s (per lane)
  int snap_step = 0;    // s            (wave-uniform: every lane refreshes at the same steps)
  int next_snap = 2;    // refresh the snapshot when the step index reaches this (2, 4, 8, ...)
  int lambda = 0;       // cycle length of this ray, 0 = not known yet
  nexec = steps;
  // |v| (1 + 1e-4): the point moves by |f| |v| per step, which is what lets a cull decision be carried
  // from one step to the next (Scene::eval_near); every 16th step the knowledge is dropped, so rounding
  // drift of p cannot accumulate beyond the slack folded into the bound (derive_constants)
  const float vn = 1.0001f * __builtin_amdgcn_sqrtf(__builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
  float move = __builtin_nanf("");
#if RM_PRIO_TILE > 0
  __builtin_amdgcn_s_setprio(0);
#endif
  for (int i = 0; i < steps; ++i) {
#if RM_PRIO_TILE > 0
    // a tile still marching by now is one of those that decide when the launch ends: its wave is issued first from
    // here on (measured, profiles/regen_probe.py: 32-primitive 8K band 11.45 -> 10.65 ms, 1080p scene 2 208 -> 204 us)
    if (i == RM_PRIO_TILE) __builtin_amdgcn_s_setprio(3);
#endif
    if (traj && live) traj_store<kSoa>(traj, traj_stride, ray, i, p);
    float f = scene.eval_near(p, (i & 15) ? move : __builtin_nanf(""));
    move = __builtin_fmaf(fabsf(f), vn, 4e-6f);
    V3 pn = step_point(p, v, f);
    const int cadence = (i < RM_EARLY_DENSE_STEPS) ? 1 : 3;
    if (early && (i & cadence) == cadence) {
      // Looked at after every second step at first and after every fourth from step RM_EARLY_DENSE_STEPS on
      // (measured: 8 -> 9.35 Grays/s, 16 -> 9.3, 32 -> 9.1, every second step throughout -> 8.8; a third
      // level, every eighth step from 64 on, bought nothing).  The candidate periods i + 1 - s are then
      // multiples of 2 or 4: a multiple of the true period serves `remaining mod lambda` just as well, and
      // a fixed point is a cycle of any length; a wave leaves at most 3 steps later than it could.
      // branch-free per-lane bookkeeping (selects, no exec-mask juggling)
      const bool fixed = same_bits(pn, p);
      // (also while a trajectory is recorded: the reverse sweep then reads p_final for every step >= nexec, which
      // is what a cycle between neighbouring floats is up to an ulp or two -- far inside the tolerance tau of its
      // converged-tail handling, march_reverse)
      const bool cyc = same_bits(pn, snap);
      const int found = fixed ? 1 : (cyc ? (i + 1 - snap_step) : 0);
      lambda = (lambda == 0) ? found : lambda;
      if constexpr (ParkF::kEnabled) {
        // a minority of the tile's rays still moving at a check step: hand them over instead of keeping 64 lanes
        // (and the whole wave's issue slots) busy for them
        const int done = i + 1;
        if (!traj && done >= RM_PARK_FIRST * pstride && done < steps && done % pstride == 0) {
          const bool unsettled = live && lambda == 0;
          const int cnt = __popcll(__ballot(unsettled));
          if (cnt > 0 && cnt <= RM_PARK_MAX_LANES) {
            const bool gone = park(done, pn, unsettled);
            if (gone) { *parked = true; lambda = 1; }          // no longer waited for; its pixel is not ours any more
          }
        }
      }
      if (__all(lambda > 0)) {
        nexec = i + 1;
        int need = (steps - (i + 1)) % lambda;    // further steps this lane still has to take
        p = pn;
        while (__any(need > 0)) {                 // < max lambda iterations; lanes freeze when done
          float g = scene.eval(p);
          V3 q = mk3(g * v.x + p.x, g * v.y + p.y, g * v.z + p.z);
          if (need > 0) p = q;
          --need;
        }
        return p;
      }
      if (i + 1 == next_snap) {                   // scalar condition; lanes that already know their
        snap = pn; snap_step = i + 1;             // period no longer look at the snapshot
        next_snap <<= 1;
      }
    }
    p = pn;
  }
  return p;
}

template <class Cfg>
__global__ void __launch_bounds__(256) k_march_fwd(RmScene sc, const void* __restrict__ pos, const void* __restrict__ dirs,
                            void* __restrict__ out, float* __restrict__ traj, int32_t* __restrict__ nexec_out,
                            int64_t n, int steps, int flags, int dt) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(sc, rm_smem, store);
  int64_t ntiles = (n + blockDim.x - 1) / blockDim.x;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t i = tile * blockDim.x + threadIdx.x;
    bool live = i < n;
    int64_t ic = live ? i : n - 1;
    int nexec;
    V3 p = march(scene, load3_t(pos, ic, dt), load3_t(dirs, ic, dt), steps, flags & RM_FLAG_EARLY_OUT, traj, n, ic, live, nexec);
    if (live) {
      store3_t(out, i, p, dt);
      if (nexec_out) nexec_out[i] = nexec;
    }
  }
}

// Reverse sweep of the march: lambda_i = lambda_{i+1} + (lambda_{i+1}.v) grad_p f(p_i),
// dL/dtheta += (lambda_{i+1}.v) df/dtheta(p_i), dL/dv += f(p_i) lambda_{i+1}.
//
// Early exit (`early`): every contribution of step i is proportional to g_i = lambda_{i+1}.v, and
// g shrinks by (1 + grad f . v) per step through the converged tail of the trajectory (0 for a
// head-on hit).  Once |g| is below the rounding-error bound of its own dot product,
// 4 eps (|l_x v_x| + |l_y v_y| + |l_z v_z|), it carries no information; the wave stops when that
// holds for all 64 rays.  The reference keeps adding such noise terms; the difference is far
// inside the 1e-4 gradient tolerance (tests: worst |grad error| unchanged at 1e-6 level).
struct NoDefer {
  RM_DEV bool operator()(int, V3, V3, bool) const { return false; }
};

#ifndef RM_BWD_INLINE_STEPS
#define RM_BWD_INLINE_STEPS 3     // moving steps a wave still walks itself before handing its active rays over
#endif

// Phase clocks of k_render_bwd (-DRM_BWD_STAMPS, profiles/bwd_phases.py): s_memtime at phase boundaries, per wave, summed
// into workspace words 8..17 at the end of the kernel.  Compiled out otherwise.
#ifdef RM_BWD_STAMPS
struct Stamps {
  unsigned long long t;
  unsigned acc[10];
  RM_DEV void start() { t = __builtin_readcyclecounter(); for (int k = 0; k < 10; ++k) acc[k] = 0; }
  RM_DEV void mark(int k) { const unsigned long long now = __builtin_readcyclecounter(); acc[k] += (unsigned)((now - t) >> 4); t = now; }
};
#define RM_STAMP(st, k) do { if (st) (st)->mark(k); } while (0)
#else
struct Stamps {};
#define RM_STAMP(st, k) do { } while (0)
#endif

// `defer(i, lambda, gv, active)`: called once, wave-wide, when more than RM_BWD_INLINE_STEPS non-converged steps
// remain; returns (per lane) whether the ray was handed to the deferred-ray kernels, which then own its outputs.
template <class SceneT, class DeferF = NoDefer, bool kSoa = false>
RM_DEV V3 march_reverse(const SceneT& scene, V3 lam, V3 v, V3 p_final, const float* traj, int64_t traj_stride,
                        int64_t ray, int nexec, int steps, bool want_gv, V3& gv, bool early, int* walked = nullptr,
                        DeferF defer = DeferF(), bool* deferred_out = nullptr, [[maybe_unused]] Stamps* stamps = nullptr) {
  if (deferred_out) *deferred_out = false;
  // `gv` by reference and a flag, not an optional pointer: a pointer that may be null pins the vector in
  // scratch memory (a load + store + vmcnt(0) per step).  The iterate of the NEXT step is fetched before
  // this step's VJP, so its HBM/L2 latency hides behind ~1000 instructions instead of stalling the wave.
  if (walked) *walked = 0;
  if (steps <= 0) return lam;               // no trajectory buffer at all in that case
  int i = steps - 1;
  auto iterate = [&](int k) { return (k < nexec) ? traj_load<kSoa>(traj, traj_stride, ray, k) : p_final; };
  auto noise = [&](float gf) {              // |g| below the rounding-error bound of its own dot product
    return fabsf(gf) <= 2.4e-7f * ((fabsf(lam.x * v.x) + fabsf(lam.y * v.y)) + fabsf(lam.z * v.z));
  };
  auto finish_frozen = [&](int k) {         // steps 0..k skipped with lambda frozen: sum_i f(p_i) = (p_{k+1} - p_0).v / |v|^2
    if (!want_gv) return;
    V3 dp = iterate(k + 1) - traj_load<kSoa>(traj, traj_stride, ray, 0);
    float sumf = ((dp.x * v.x + dp.y * v.y) + dp.z * v.z) / ((v.x * v.x + v.y * v.y) + v.z * v.z);
    gv = gv + sumf * lam;
  };
#ifndef RM_BWD_NO_TAIL
  // Converged tail (`early` only).  Once the march has settled, every remaining iterate is the same point up to
  // the last few ulps (a bitwise fixed point, a 2-4 cycle between neighbouring floats, or a ray creeping 1e-7
  // per step along a wall), so grad f and df/dtheta there are the same for all those steps: the recursion
  //   g_i = lambda.v,  lambda += g_i n,  dL/dtheta += g_i df/dtheta(p)        (n = grad_p f at the anchor)
  // needs ONE point-gradient evaluation for n, a few flops per step, and ONE parameter VJP with the summed
  // upstream G = sum g_i at the end -- instead of a full VJP per step (measured on config 4: the reverse sweep
  // walked 24.6 steps per wave tile on average, nearly all of them inside this tail).  The anchor is renewed
  // when some ray of the wave has moved more than tau = 1e-6 max(1, |p|) away from it; arguments of the VJP
  // differ from the reference's by <= tau, far inside the 1e-4 gradient tolerance (tests: f5_backward).
  // Every iterate below `nexec` is a load, and a load the next wave-wide vote depends on costs its full latency (~2 us
  // behind a 200 MB trajectory: 67 % of this kernel's wave-cycles were s_waitcnt, profiles/r03_stalls.txt).  The loop
  // therefore keeps a WINDOW of the next eight iterates down, fetched together -- before the anchor's point-gradient
  // evaluation, whose ~600 instructions hide them -- and consumed with constant indices.
  V3 cur = (early && i >= 0) ? iterate(i) : p_final;           // the iterate of step i
  while (early && i >= 0) {
    const V3 anchor = cur;
    const float tau = 1e-6f * fmaxf(1.0f, fmaxf(fabsf(anchor.x), fmaxf(fabsf(anchor.y), fabsf(anchor.z))));
    auto near_anchor = [&](V3 q) {
      return fmaxf(fabsf(q.x - anchor.x), fmaxf(fabsf(q.y - anchor.y), fabsf(q.z - anchor.z))) <= tau;
    };
    V3 win[8];                                                   // iterates of steps i-1 .. i-8 (the anchor where there is none)
    auto fill = [&](int top) {
#pragma unroll
      for (int j = 0; j < 8; ++j) win[j] = (top - j >= 0) ? iterate(top - j) : anchor;
    };
    fill(i - 1);
    // how many of the next steps down stay at the anchor (for every ray of the wave)?  at least two, or the
    // plain per-step VJP below is cheaper
    if (i < 1 || !__all(near_anchor(win[0]))) break;
    RM_STAMP(stamps, 2);
    float f0;
    const V3 n = scene.vjp_point(anchor, 1.0f, &f0);
    RM_STAMP(stamps, 3);
    float G = 0.0f;
    bool done = false, more = true;
    V3 p_i = anchor;
    while (more) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (i < 0 || !__all(near_anchor(p_i))) { more = false; break; }
        const float gf = (lam.x * v.x + lam.y * v.y) + lam.z * v.z;
        if (__all(noise(gf))) { done = true; more = false; break; }
        G = G + gf;
        if (want_gv) gv = gv + f0 * lam;
        lam = lam + mk3(gf * n.x, gf * n.y, gf * n.z);
        --i;
        p_i = win[j];                                           // the iterate of the new step i (checked above when i < 0)
      }
      if (more) fill(i - 1);                                    // p_i = win[7] is step i; the window moves on
    }
    cur = p_i;
    RM_STAMP(stamps, 4);
    // parameter gradients of the whole run at once: the reverse pass alone, on the tape the point-gradient pass left
    // (no scene evaluation since: the loop above only loaded iterates)
    if (__any(G != 0.0f)) scene.vjp_replay(anchor, G);
    RM_STAMP(stamps, 5);
    if (walked) *walked += 2;
    if (done) { finish_frozen(i); return lam; }
  }
  if (i < 0) return lam;
#endif
  // Rays that are still travelling (grazing a surface, threading a blend region) need a VJP at every one of
  // their remaining steps, and a wave that walks them one after the other IS the critical path of the whole
  // launch (config 4: 7 % of the wave tiles walked all 64 steps, with ~14 of their 64 lanes still active, while
  // the rest of the GPU idled).  Such rays are handed to k_bwd_hard_n / _a / _b, which evaluate all their
  // (ray, step) pairs in parallel; lanes whose adjoint component along the ray is already noise are finished.
  if (early && i + 1 > RM_BWD_INLINE_STEPS) {
    const float gf0 = (lam.x * v.x + lam.y * v.y) + lam.z * v.z;
    const bool active = !noise(gf0);
    if (!__any(active)) { finish_frozen(i); return lam; }
    const bool deferred = defer(i, lam, gv, active);
    if (deferred) {
      if (deferred_out) *deferred_out = true;
      lam = mk3(0.0f, 0.0f, 0.0f);                 // contributes nothing below; its outputs come from k_bwd_hard_a
    }
    if (!__any(active && !deferred)) {               // nobody left to walk here
      if (!deferred) finish_frozen(i);
      return lam;
    }
  }
  V3 p_next = iterate(i);
  for (; i >= 0; --i) {
    float gf = (lam.x * v.x + lam.y * v.y) + lam.z * v.z;
    if (early && __all(noise(gf))) { finish_frozen(i); break; }
    V3 p = p_next;
    if (i > 0) p_next = iterate(i - 1);
    float f;
    V3 gp = scene.vjp(p, gf, &f);
    if (want_gv) gv = gv + f * lam;
    lam = lam + gp;
    if (walked) *walked += 1;
  }
  return lam;
}

template <class Cfg>
__global__ void __launch_bounds__(256) k_march_bwd(RmScene sc, const float* __restrict__ dirs, const float* __restrict__ traj,
                            const int32_t* __restrict__ nexec, const float* __restrict__ gout,
                            float* __restrict__ gpos, float* __restrict__ gdirs, float* __restrict__ partials,
                            int64_t n, int steps) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(sc, rm_smem, store, true);
  const int n_acc = Cfg::n_acc(sc);
  zero_accumulators<Cfg>(scene, n_acc);
  int64_t ntiles = (n + blockDim.x - 1) / blockDim.x;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t i = tile * blockDim.x + threadIdx.x;
    bool live = i < n;
    int64_t ic = live ? i : n - 1;
    V3 lam = live ? load3(gout, ic) : mk3(0.0f, 0.0f, 0.0f);
    V3 v = load3(dirs, ic);
    V3 gv = mk3(0.0f, 0.0f, 0.0f);
    // With a trajectory recorded, every iterate from nexec-1 on equals the last stored one up to the ulp or two
    // of a cycle between neighbouring floats (exactly, for a fixed point).
    int ne = nexec ? nexec[ic] : steps;
    V3 pf = (ne > 0) ? load3(traj + 3 * (int64_t)(ne - 1) * n, ic) : mk3(0.0f, 0.0f, 0.0f);
    lam = march_reverse(scene, lam, v, pf, traj, n, ic, ne, steps, gdirs != nullptr, gv, false);
    if (live) {
      if (gpos) store3(gpos, i, lam);
      if (gdirs) store3(gdirs, i, gv);
    }
  }
  flush_accumulators<Cfg>(scene, n_acc, partials, rm_smem + ((sc.n_params + sc.n_derived + 3) & ~3));
}

// ---------------------------------------------------------------------------
// normals
// ---------------------------------------------------------------------------
template <class Cfg>
__global__ void __launch_bounds__(256) k_normals_fwd(RmScene sc, RmTetra tetra, const void* __restrict__ pts, void* __restrict__ nrm,
                              void* __restrict__ lap, int64_t n, int dt) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(sc, rm_smem, store);
  Tetra T = load_tetra(tetra);
  int64_t ntiles = (n + blockDim.x - 1) / blockDim.x;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t i = tile * blockDim.x + threadIdx.x;
    bool live = i < n;
    int64_t ic = live ? i : n - 1;
    V3 p = load3_t(pts, ic, dt);
    float c = scene.eval(p);
    V3 nn; float ll;
    normals_forward(scene, T, p, c, nn, ll);
    if (live) {
      store3_t(nrm, i, nn, dt);
      st_t(lap, i, ll, dt);
    }
  }
}

// VJP of normals_forward.  u: the un-normalised normal of the forward pass, gn: dL/dn, gl: dL/dlap.  Returns dL/dp.
template <class SceneT>
RM_DEV V3 normals_backward(const SceneT& sc, const Tetra& T, V3 p, V3 u, V3 gn, float gl, bool need_lap) {
  float nu = norm3(u);
  V3 n = mk3(u.x / nu, u.y / nu, u.z / nu);
  // n = u / |u|  ->  g_u = (g_n - n (n.g_n)) / |u|
  float ng = (n.x * gn.x + n.y * gn.y) + n.z * gn.z;
  V3 gu = mk3((gn.x - n.x * ng) / nu, (gn.y - n.y * ng) / nu, (gn.z - n.z * ng) / nu);
  float g1 = (T.inv[0] * gu.x + T.inv[3] * gu.y) + T.inv[6] * gu.z;
  float g2 = (T.inv[1] * gu.x + T.inv[4] * gu.y) + T.inv[7] * gu.z;
  float g3 = (T.inv[2] * gu.x + T.inv[5] * gu.y) + T.inv[8] * gu.z;
  float g0 = -((g1 + g2) + g3);
  V3 gp = mk3(0.0f, 0.0f, 0.0f);
  if (need_lap) {  // lap = (f(p) - mean(taps)) * scale
    float gs = gl * T.lap_scale;
    float gm = -gs / 4.0f;
    g0 += gm; g1 += gm; g2 += gm; g3 += gm;
    gp = gp + sc.vjp(p, gs);
  }
  // one rolled loop over the four taps: a single inlined VJP body instead of four (the backward frame
  // kernel drops from 185 to fewer VGPRs and a quarter of the code)
  const float ax = T.o[0].x, ay = T.o[0].y, az = T.o[0].z, bx = T.o[1].x, by = T.o[1].y, bz = T.o[1].z;
  const float cx = T.o[2].x, cy = T.o[2].y, cz = T.o[2].z, dx = T.o[3].x, dy = T.o[3].y, dz = T.o[3].z;
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const V3 ok = mk3((k == 0) ? ax : ((k == 1) ? bx : ((k == 2) ? cx : dx)),
                      (k == 0) ? ay : ((k == 1) ? by : ((k == 2) ? cy : dy)),
                      (k == 0) ? az : ((k == 1) ? bz : ((k == 2) ? cz : dz)));
    const float gk = (k == 0) ? g0 : ((k == 1) ? g1 : ((k == 2) ? g2 : g3));
    gp = gp + sc.vjp(p + ok, gk);
  }
  return gp;
}

template <class Cfg>
__global__ void __launch_bounds__(256) k_normals_bwd(RmScene sc, RmTetra tetra, const float* __restrict__ pts,
                              const float* __restrict__ gn, const float* __restrict__ gl,
                              float* __restrict__ gpts, float* __restrict__ partials, int64_t n) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(sc, rm_smem, store, true);
  const int n_acc = Cfg::n_acc(sc);
  zero_accumulators<Cfg>(scene, n_acc);
  Tetra T = load_tetra(tetra);
  int64_t ntiles = (n + blockDim.x - 1) / blockDim.x;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t i = tile * blockDim.x + threadIdx.x;
    bool live = i < n;
    int64_t ic = live ? i : n - 1;
    V3 g = (gn && live) ? load3(gn, ic) : mk3(0.0f, 0.0f, 0.0f);
    float gls = (gl && live) ? gl[ic] : 0.0f;
    const V3 pt = load3(pts, ic);
    V3 nn, uu; float ll;
    normals_forward(scene, T, pt, 0.0f, nn, ll, &uu);
    V3 gp = normals_backward(scene, T, pt, uu, g, gls, gl != nullptr);
    if (live && gpts) store3(gpts, i, gp);
  }
  flush_accumulators<Cfg>(scene, n_acc, partials, rm_smem + ((sc.n_params + sc.n_derived + 3) & ~3));
}

// ---------------------------------------------------------------------------
// fused frame: camera -> march -> distance -> normals -> shader
// ---------------------------------------------------------------------------
struct RenderArgs {
  RmScene scene;
  RmCamera cam;
  RmTetra tetra;
  const void* orientation;   // [N,4]  element type cam.dtype
  const void* translation;   // [N,3]
  void* image;               // [N,rows,W,3] of image_dtype
  float* first_pass;         // modes 1, 2, 5: fp32 [N,rows,W,3] un-normalised values
  float* p_final;            // nullable
  float* traj;               // nullable [wave tile][step][3][64]  (traj_store<true>: slot = tile * 64 + lane)
  int32_t* nexec;            // nullable
  float* normal_u;           // nullable [R,3]: the un-normalised normal of the final point (training frames: the backward
                             // kernel then needs neither the four tap evaluations of the shader VJP nor those of the normalisation VJP)
  uint32_t* minmax;          // nullable
  const void* cmap;          // nullable, [cmap_size,3] of cmap_dtype
  int32_t cmap_size, cmap_dtype, image_dtype;
  int32_t mode, degree, steps, row_begin, row_end, flags;
  const int32_t* tile_order; // nullable: position in the dealing order -> wave tile (longest-first schedules)
  int32_t* tile_cost;        // nullable out: march steps executed by the wave of each tile
  // backward only
  const float* grad_image;
  float* partials;
  float* grad_pos;           // nullable [R,3]: dL/d(ray origin)   (feeds rm_camera_backward)
  float* grad_dirs;          // nullable [R,3]: dL/d(ray direction)
  float* grad_qdir;          // nullable [R,4]: dL/d(orientation) through the SHADER's own use of the pose (modes 3, 6, 7)
  // rays parked by the forward kernel for k_render_parked (nullable / 0)
  int32_t* park_ray;         // [lists * shards][seg] band-output index
  float* park_p;             // [lists * shards][seg][3] iterate after the check step
  int32_t park_seg;          // rays per (list, shard) segment
  // deferred rays of the reverse sweep (k_bwd_hard_*), all nullable / 0
  int32_t* hard_ray;         // [cap] band-output index of the ray
  int32_t* hard_slot;        // [cap] its trajectory slot (wave tile * 64 + lane)
  int32_t* hard_step;        // [cap] highest step index still to be walked
  float* hard_state;         // [cap][8] lambda(3), dL/dv so far(3)
  float* hard_n;             // [steps][cap][4] grad_p f(p_s) for unit upstream, f(p_s)
  float* hard_g;             // [steps][cap] upstream g_s
  float* hard_p;             // [steps][cap][4] the iterate p_s itself (k_bwd_hard_b reads it by pair code: no ray -> nexec -> trajectory chain)
  uint32_t* hard_pairs;      // [steps * cap] (ray slot << 11 | step) of every pair with g != 0, densely packed
  int32_t hard_cap;
};

#define RM_WORK_HARD_COUNT 32      // workspace word (own 128-B line): rays deferred by k_render_bwd
#define RM_WORK_HARD_PAIRS 33      // workspace word: (ray, step) pairs listed by k_bwd_hard_a for k_bwd_hard_b
#define RM_HARD_STEP_BITS 11       // steps < 2048, slots < 2^21 (checked on the host)

// Work decomposition of a frame: a *wave tile* is 64 rays handled by one wavefront --
// an 8x8 pixel tile (RM_FLAG_TILE8X8, better convergence coherence for the wave-uniform
// early-out) or 64 consecutive pixels of the row-major band.
RM_DEV int64_t wave_tiles(const RenderArgs& a) {
  const int W = a.cam.width, rows = a.row_end - a.row_begin;
  if (a.flags & RM_FLAG_TILE8X8) return (int64_t)a.cam.num_cameras * ((W + 7) >> 3) * ((rows + 7) >> 3);
  return ((int64_t)a.cam.num_cameras * rows * W + 63) >> 6;
}

RM_DEV bool ray_of_tile_lane(const RenderArgs& a, int64_t wave_tile, int lane, int& cam, int& row, int& col) {
  const int W = a.cam.width, rows = a.row_end - a.row_begin;
  if (a.flags & RM_FLAG_TILE8X8) {
    const int tw = (W + 7) >> 3, th = (rows + 7) >> 3;
    int64_t per_cam = (int64_t)tw * th;
    cam = (int)(wave_tile / per_cam);
    int64_t t = wave_tile - (int64_t)cam * per_cam;
    int ty = (int)(t / tw), tx = (int)(t - (int64_t)ty * tw);
    row = ty * 8 + (lane >> 3);
    col = tx * 8 + (lane & 7);
    return cam < a.cam.num_cameras && row < rows && col < W;
  }
  int64_t li = wave_tile * 64 + lane;
  int64_t per_cam = (int64_t)rows * W;
  cam = (int)(li / per_cam);
  int64_t r = li - (int64_t)cam * per_cam;
  row = (int)(r / W);
  col = (int)(r - (int64_t)row * W);
  return cam < a.cam.num_cameras;
}

RM_DEV bool ray_of_lane(const RenderArgs& a, int64_t wave_tile, int& cam, int& row, int& col) {
  return ray_of_tile_lane(a, wave_tile, threadIdx.x & 63, cam, row, col);
}

// Dynamic tile distribution.  The bit-exact early-out makes tile cost vary ~6x (config 2: 30% of
// the wave tiles -- rays sliding along the side walls -- need all 128 steps, the rest ~20) and
// a static stride of 4 tiles per wave ends with a tail ~2.4x the mean.  A single atomic counter
// is no cure: same-address returning atomics retire at 12-16 ns each chip-wide (measured,
// profiles/micro/atomic_bench.hip), i.e. 0.5 ms for the 32k tiles of one frame.  So the tiles
// are dealt round-robin into RM_WORK_QUEUES counters on separate 128-B lines; a wave drains
// its home queue and then steals from the next non-empty one.  Emptiness of all queues is read
// with ONE 64-lane agent-scope load (lane l reads counter l).  Counters only grow, every tile
// index is handed out exactly once, and a wave leaves when every counter has passed its queue's
// size, so the grid always drains.
RM_DEV uint32_t queue_size(int64_t ntiles, int q) {
  return (q < ntiles) ? (uint32_t)((ntiles - q + RM_WORK_QUEUES - 1) / RM_WORK_QUEUES) : 0u;
}

// Every wave's FIRST tile is static: wave w takes position w/64 of its home queue w%64, so the
// counters logically start at `first_positions(q)` (the number of waves whose home is q) and the
// burst of one atomic per wave at kernel start -- the most contended moment -- never happens.
RM_DEV uint32_t first_positions(int q) {
  const int total_waves = gridDim.x * (blockDim.x >> 6);
  return (q < total_waves) ? (uint32_t)((total_waves - q + RM_WORK_QUEUES - 1) / RM_WORK_QUEUES) : 0u;
}

RM_DEV int64_t grab_wave_tile(uint32_t* work, int64_t ntiles, int& q) {
  const int lane = threadIdx.x & 63;
  uint32_t* ctr = work + RM_WORK_QUEUE_BASE;
  for (int attempt = 0; attempt < 4 * RM_WORK_QUEUES; ++attempt) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&ctr[q * RM_WORK_QUEUE_STRIDE], 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) + first_positions(q);
    if (t < queue_size(ntiles, q)) return (int64_t)q + (int64_t)t * RM_WORK_QUEUES;
    // home queue is empty: look at all of them at once
    uint32_t c = __hip_atomic_load(&ctr[lane * RM_WORK_QUEUE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long avail = __ballot(c + first_positions(lane) < queue_size(ntiles, lane));
    if (avail == 0ull) return -1;
    // steal from the first non-empty queue after a per-wave pseudo-random start (if every wave of an
    // emptied queue moved to the SAME neighbour they would travel round the ring as a convoy)
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int r = (int)(((wave * 2654435761u) >> 20) + attempt * 17) & 63;
    unsigned long long rot = (r == 0) ? avail : ((avail >> r) | (avail << (64 - r)));
    q = (r + __builtin_ctzll(rot)) & 63;
  }
  // not reached in practice; finish with a definitive sweep so no tile can be dropped
  for (int k = 0; k < RM_WORK_QUEUES; ++k) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&ctr[k * RM_WORK_QUEUE_STRIDE], 1u);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) + first_positions(k);
    if (t < queue_size(ntiles, k)) { q = k; return (int64_t)k + (int64_t)t * RM_WORK_QUEUES; }
  }
  return -1;
}

struct TileCursor {
  int64_t tile;
  int q;
};

RM_DEV TileCursor first_wave_tile(const RenderArgs& a, int64_t ntiles) {
  TileCursor c;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  c.q = (int)(wave & (RM_WORK_QUEUES - 1));
  if (a.minmax && (a.flags & RM_FLAG_DYNAMIC_TILES)) {
    // static first tile: position wave/64 of the home queue (tile index == wave when it exists)
    c.tile = (wave < ntiles) ? wave : ntiles;
  } else {
    c.tile = wave;
  }
  return c;
}

RM_DEV void next_wave_tile(const RenderArgs& a, int64_t ntiles, TileCursor& c) {
  if (a.minmax && (a.flags & RM_FLAG_DYNAMIC_TILES)) {
    c.tile = grab_wave_tile(a.minmax, ntiles, c.q);
    if (c.tile < 0) c.tile = ntiles;
  } else {
    c.tile += (int64_t)gridDim.x * (blockDim.x >> 6);
  }
}

RM_DEV Pose load_pose(const void* orientation, const void* translation, int cam, int dt = RM_DTYPE_F32) {
  Pose ps;
  ps.w = ld_t(orientation, 4 * cam, dt);
  ps.qv = mk3(ld_t(orientation, 4 * cam + 1, dt), ld_t(orientation, 4 * cam + 2, dt), ld_t(orientation, 4 * cam + 3, dt));
  ps.t = load3_t(translation, cam, dt);
  return ps;
}

// One shaded pixel: rgb, or -- tangent / spin shaders -- a brightness and a colormap row whose product is taken
// in the colormap's own type when the pixel is stored (store_shaded).
struct Shaded {
  V3 rgb;
  float bright;
  int idx;      // >= 0: colour = bright * cmap[idx]
};
RM_DEV Shaded plain(V3 rgb) { return Shaded{rgb, 0.0f, -1}; }

// angle_colouring / domain_colouring (shader.py:92-118)
RM_DEV Shaded domain_colour(float re, float im, int size, int degree) {
  const float tau = 6.283185307179586f;
  float x = (((rm_atan2(im, re) / tau) + 0.5f) * (float)degree) * (float)size;
  long long idx = (long long)floorf(x);
  long long m = idx % size;
  if (m < 0) m += size;
  float bright = rm_sqrt(re * re + im * im);
  if (!(x == x)) { m = 0; bright = x; }  // NaN normal: propagate NaN instead of indexing with it
  return Shaded{mk3(0.0f, 0.0f, 0.0f), bright, (int)m};
}

// brightness.mul(colours) (shader.py:118): fp32 brightness times a float64 colormap row is a float64 product
// (type promotion), times an fp32 / fp16 row a product in that type.
// RM_DTYPE_RGBA_F32: what `images.mean(0).float()` padded with alpha 1 holds for ONE camera -- the value rounded to the
// type the [N,H,W,3] image would have had (`round_dt`: the module's dtype, or float64 for a float64 colormap product),
// then to fp32
RM_DEV void store_rgba(void* image, int64_t li, V3 c, int round_dt) {
  if (round_dt == RM_DTYPE_F16) c = mk3((float)(_Float16)c.x, (float)(_Float16)c.y, (float)(_Float16)c.z);
  *reinterpret_cast<float4*>(static_cast<float*>(image) + 4 * li) = make_float4(c.x, c.y, c.z, 1.0f);
}

RM_DEV void store_shaded(void* image, int image_dt, int64_t li, const Shaded& s, const void* cmap, int cmap_dt, int round_dt = RM_DTYPE_F32) {
  if (s.idx < 0) {
    if (image_dt == RM_DTYPE_F32) store3(static_cast<float*>(image), li, s.rgb);
    else if (image_dt == RM_DTYPE_RGBA_F32) store_rgba(image, li, s.rgb, round_dt);
    else store3_t(image, li, s.rgb, image_dt);
    return;
  }
  if (cmap_dt == RM_DTYPE_F64) {
    const double* c = static_cast<const double*>(cmap) + 3 * (int64_t)s.idx;
    const double b = (double)s.bright;
    const double r0 = b * c[0], r1 = b * c[1], r2 = b * c[2];
    if (image_dt == RM_DTYPE_F64) {
      double* o = static_cast<double*>(image) + 3 * li;
      o[0] = r0; o[1] = r1; o[2] = r2;
    } else if (image_dt == RM_DTYPE_RGBA_F32) {
      store_rgba(image, li, mk3((float)r0, (float)r1, (float)r2), RM_DTYPE_F32);     // float64 image .float()
    } else {
      store3_t(image, li, mk3((float)r0, (float)r1, (float)r2), image_dt);
    }
    return;
  }
  const V3 c = load3_t(cmap, s.idx, cmap_dt);
  // (brightness in the module's type times a colormap row of cmap_dt: torch's type promotion -- fp16 x fp16 stays fp16)
  const V3 prod = mk3(s.bright * c.x, s.bright * c.y, s.bright * c.z);
  if (image_dt == RM_DTYPE_RGBA_F32) store_rgba(image, li, prod, (round_dt == RM_DTYPE_F16 && cmap_dt == RM_DTYPE_F16) ? RM_DTYPE_F16 : RM_DTYPE_F32);
  else store3_t(image, li, prod, image_dt);
}

struct ShadeIn {
  V3 o, v, p, n;      // pixel (ray origin) position, ray direction, surface point, surface normal
  float lap, dist;
  float qw; V3 qv;    // camera orientation
  V3 col2;            // pixel_frames[..., 2]
};

// One pixel of Shader.forward (shader.py:190-263).  Modes 1, 2, 5 return the
// un-normalised value; rm_shade_finish applies the global min/max.
RM_DEV Shaded shade_pixel(int mode, const ShadeIn& s, int cmap_size, int degree) {
  switch (mode) {
    case RM_MODE_LAMBERTIAN: {  // shader.py:16-20
      float c = t_clamp(-dot_seq(s.v, s.n), 0.0f, 1.0f);
      return plain(mk3(c, c, c));
    }
    case RM_MODE_DISTANCE: {    // shader.py:27-33
      float l = rm_log(t_clamp(norm3(s.o - s.p), 1e-2f, __builtin_inff()));
      return plain(mk3(l, l, l));
    }
    case RM_MODE_PROXIMITY: {   // shader.py:45-50
      float l = rm_log(t_clamp(s.dist, 1e-2f, __builtin_inff()));
      return plain(mk3(l, l, l));
    }
    case RM_MODE_VIGNETTE: {    // shader.py:62-66
      float d = dot_seq(s.v, s.col2);
      float c = (d * d) * d;
      return plain(mk3(c, c, c));
    }
    case RM_MODE_NORMAL:        // shader.py:73-74
      return plain(mk3(t_clamp(fabsf(s.n.x), 0.0f, 1.0f), t_clamp(fabsf(s.n.y), 0.0f, 1.0f),
                       t_clamp(fabsf(s.n.z), 0.0f, 1.0f)));
    case RM_MODE_LAPLACIAN:     // shader.py:81-89
      return plain(mk3(s.lap, s.lap, s.lap));
    case RM_MODE_TANGENT: {     // shader.py:125-150
      float c = dot_seq(s.n, s.v);
      V3 tg = mk3((c * s.v.x) * -1.0f + s.n.x, (c * s.v.y) * -1.0f + s.n.y, (c * s.v.z) * -1.0f + s.n.z);
      V3 pr = qrot(tg, s.qw, neg(s.qv));
      return domain_colour(pr.x, pr.y, cmap_size, degree);
    }
    default: {                  // RM_MODE_SPIN, shader.py:157-171: (0,n) * conj(q)
      float q0 = s.qw, q1 = -s.qv.x, q2 = -s.qv.y, q3 = -s.qv.z;
      float p0 = 0.0f, p1 = s.n.x, p2 = s.n.y, p3 = s.n.z;
      float r0 = p0 * q0 - ((p1 * q1 + p2 * q2) + p3 * q3);
      float r1 = ((p0 * q1 + p1 * q0) + p2 * q3) - p3 * q2;
      float r2 = ((p0 * q2 + p2 * q0) + p3 * q1) - p1 * q3;
      float r3 = ((p0 * q3 + p1 * q2) + p3 * q0) - p2 * q1;
      float re = r0 * r0 - ((r1 * r1 + r2 * r2) + r3 * r3);
      float im = (norm3(mk3(r1, r2, r3)) * r0) * 2.0f;
      return domain_colour(im, re, cmap_size, degree);   // (imag, real) swapped as in the reference
    }
  }
}

// fold a wave's running min/max (+NaN flag) into one of the workspace's partial triples (include/rm_abi.h): every
// wave of a frame folding into the SAME three words serialises ~15000 atomics at one L2 line -- 60 us of the 470 us
// distance-shader frame at 1080p
RM_DEV void fold_minmax(uint32_t* minmax, float lo, float hi, bool saw_nan) {
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  unsigned long long nanmask = __ballot(saw_nan);
  if ((threadIdx.x & 63) == 0) {
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    uint32_t* slot = minmax + RM_WORK_MM_BASE + 32 * (wave % RM_WORK_MM_SLOTS);
    atomicMin(&slot[0], f2ord(lo));
    atomicMax(&slot[1], f2ord(hi));
    if (nanmask) atomicOr(&slot[2], 1u);
  }
}

// the global {min, max, NaN flag}: words 0-2 combined with the partial triples; every lane of the wave gets it
RM_DEV void load_minmax(const uint32_t* __restrict__ mm, float& lo, float& hi) {
  const int lane = threadIdx.x & 63;
  const uint32_t* slot = mm + RM_WORK_MM_BASE + 32 * (lane % RM_WORK_MM_SLOTS);
  uint32_t a = slot[0], b = slot[1], c = slot[2];
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t a2 = __shfl_xor(a, o, 64), b2 = __shfl_xor(b, o, 64), c2 = __shfl_xor(c, o, 64);
    a = a2 < a ? a2 : a; b = b2 > b ? b2 : b; c |= c2;
  }
  a = mm[0] < a ? mm[0] : a; b = mm[1] > b ? mm[1] : b; c |= mm[2];
  lo = ord2f(a); hi = ord2f(b);
  if (c) { lo = __builtin_nanf(""); hi = lo; }
}

// Standalone Shader.forward over tensors (any of the inputs a mode does not read may be null).
struct ShadeArgs {
  const float *px, *orientation, *frames, *dirs, *coords, *normals, *lap, *dist;
  void* image;
  uint32_t* minmax;
  const void* cmap;
  int32_t cmap_size, mode, degree;
  int64_t n, per_camera;
  int32_t image_dtype, cmap_dtype;
};

__global__ void k_shade_fwd(ShadeArgs a) {
  float lo = __builtin_inff(), hi = -__builtin_inff();
  bool saw_nan = false;
  const V3 z = mk3(0.0f, 0.0f, 0.0f);
  int64_t ntiles = (a.n + blockDim.x - 1) / blockDim.x;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int64_t i = tile * blockDim.x + threadIdx.x;
    if (i >= a.n) continue;
    int cam = (int)(i / a.per_camera);
    ShadeIn s;
    s.o = a.px ? load3(a.px, i) : z;
    s.v = a.dirs ? load3(a.dirs, i) : z;
    s.p = a.coords ? load3(a.coords, i) : z;
    s.n = a.normals ? load3(a.normals, i) : z;
    s.lap = a.lap ? a.lap[i] : 0.0f;
    s.dist = a.dist ? a.dist[i] : 0.0f;
    s.qw = a.orientation ? a.orientation[4 * cam] : 1.0f;
    s.qv = a.orientation ? mk3(a.orientation[4 * cam + 1], a.orientation[4 * cam + 2], a.orientation[4 * cam + 3]) : z;
    s.col2 = a.frames ? mk3(a.frames[9 * cam + 2], a.frames[9 * cam + 5], a.frames[9 * cam + 8]) : z;
    const Shaded sh = shade_pixel(a.mode, s, a.cmap_size, a.degree);
    const V3 out = sh.rgb;
    store_shaded(a.image, a.image_dtype, i, sh, a.cmap, a.cmap_dtype);
    if (a.mode == RM_MODE_DISTANCE || a.mode == RM_MODE_PROXIMITY) {
      saw_nan |= (out.x != out.x);
      lo = fminf(lo, out.x); hi = fmaxf(hi, out.x);
    } else if (a.mode == RM_MODE_LAPLACIAN) {
      saw_nan |= (out.x != out.x);
      hi = fmaxf(hi, fabsf(out.x));
    }
  }
  if (a.minmax && (a.mode == RM_MODE_DISTANCE || a.mode == RM_MODE_PROXIMITY || a.mode == RM_MODE_LAPLACIAN))
    fold_minmax(a.minmax, lo, hi, saw_nan);
}

// ---- per-tile pieces of the frame kernel ------------------------------------------------------------
struct TileRays {
  V3 o, v;          // world-frame ray of this lane
  Pose ps;
  int64_t li;       // index in the band outputs
  bool live;
};

// camera-frame ray + pose of one lane, element type T: every load is issued before the first use, so one
// memory latency is paid per tile (a typed load behind its own dtype branch would wait separately each time)
template <class T>
RM_DEV void load_ray_and_pose(const RenderArgs& a, int64_t gi, int cam, V3& o, V3& v, Pose& ps) {
  const T* rp = static_cast<const T*>(a.cam.ray_positions) + 3 * gi;
  const T* rd = static_cast<const T*>(a.cam.ray_directions) + 3 * gi;
  const T* q = static_cast<const T*>(a.orientation) + 4 * cam;
  const T* t = static_cast<const T*>(a.translation) + 3 * cam;
  const T o0 = rp[0], o1 = rp[1], o2 = rp[2], v0 = rd[0], v1 = rd[1], v2 = rd[2];
  const T q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], t0 = t[0], t1 = t[1], t2 = t[2];
  o = mk3((float)o0, (float)o1, (float)o2);
  v = mk3((float)v0, (float)v1, (float)v2);
  ps.w = (float)q0; ps.qv = mk3((float)q1, (float)q2, (float)q3); ps.t = mk3((float)t0, (float)t1, (float)t2);
}

RM_DEV TileRays load_tile_rays(const RenderArgs& a, int64_t tile) {
  const int W = a.cam.width, H = a.cam.height, rows = a.row_end - a.row_begin;
  TileRays r;
  int cam, row, col;
  r.live = ray_of_lane(a, tile, cam, row, col);
  if (!r.live) { cam = 0; row = 0; col = 0; }
  r.li = ((int64_t)cam * rows + row) * W + col;                          // index in the band outputs
  int64_t gi = ((int64_t)cam * H + (row + a.row_begin)) * W + col;       // index in the camera buffers
  V3 o, v;
  if (a.cam.dtype == RM_DTYPE_F16) load_ray_and_pose<_Float16>(a, gi, cam, o, v, r.ps);
  else load_ray_and_pose<float>(a, gi, cam, o, v, r.ps);
  // PinholeCamera.forward (ray_marching.py:58-62)
  r.o = qrot(o, r.ps.w, r.ps.qv) + r.ps.t;
  r.v = qrot(v, r.ps.w, r.ps.qv);
  return r;
}

struct MinMaxAcc {
  float lo, hi;
  bool saw_nan;
};

// distance, normals / Laplacian, shader, stores (control.py:244-257)
// kAux = false: a frame that asked for none of the per-ray outputs p_final / nexec / normal_u (plain inference): the stores
// and their kernel arguments are compiled out
template <class SceneT, bool kAux = true>
RM_DEV void finish_tile(const RenderArgs& a, const SceneT& scene, const Tetra& T, const TileRays& r, V3 p, int nexec,
                        MinMaxAcc& mm, bool store_p = true) {
  const int mode = a.mode;
  // scene(p) (control.py:244) is only looked at by the proximity shader and, as the centre tap, by the Laplacian:
  // the other six modes skip the evaluation -- a fifth of this epilogue's scene evaluations
  float dist = 0.0f;
  if (mode == RM_MODE_PROXIMITY || mode == RM_MODE_LAPLACIAN) dist = scene.eval(p);
  V3 n = mk3(0.0f, 0.0f, 0.0f), u = n;
  float lap = 0.0f;
  if (mode == RM_MODE_LAMBERTIAN || mode >= RM_MODE_NORMAL) normals_forward(scene, T, p, dist, n, lap, kAux ? &u : nullptr);
  if constexpr (kAux) {
    if (a.normal_u && r.live) store3(a.normal_u, r.li, u);
  }
  ShadeIn si;
  si.o = r.o; si.v = r.v; si.p = p; si.n = n; si.lap = lap; si.dist = dist;
  si.qw = r.ps.w; si.qv = r.ps.qv;
  {   // third column of the camera rotation (QuaternionToSO3, quaternion.py:114-124)
    float w = r.ps.w, x = r.ps.qv.x, y = r.ps.qv.y, z = r.ps.qv.z;
    si.col2 = mk3(2.0f * (w * y + x * z), 2.0f * (y * z - w * x), ((w * w - x * x) - y * y) + z * z);
  }
  const Shaded sh = shade_pixel(mode, si, a.cmap_size, a.degree);
  const V3 out = sh.rgb;
  const bool global = (mode == RM_MODE_DISTANCE || mode == RM_MODE_PROXIMITY || mode == RM_MODE_LAPLACIAN);
  if (r.live) {
    if (global) store3(a.first_pass, r.li, out);          // rm_shade_finish normalises into `image`
    else store_shaded(a.image, a.image_dtype, r.li, sh, a.cmap, a.cmap_dtype, a.cam.dtype);
    if constexpr (kAux) {
      if (store_p && a.p_final) store3(a.p_final, r.li, p);
      if (a.nexec) a.nexec[r.li] = nexec;
    }
    if (mode == RM_MODE_DISTANCE || mode == RM_MODE_PROXIMITY) {
      mm.saw_nan |= (out.x != out.x);
      mm.lo = fminf(mm.lo, out.x); mm.hi = fmaxf(mm.hi, out.x);
    } else if (mode == RM_MODE_LAPLACIAN) {
      mm.saw_nan |= (lap != lap);
      mm.hi = fmaxf(mm.hi, fabsf(lap));
    }
  }
}

// puts the unsettled lanes of a wave on the list of this check step (one returning atomic per wave, on one of
// RM_PARK_SHARDS counters of the list)
struct ParkToList {
#ifdef RM_PARKING
  static constexpr bool kEnabled = true;
#else
  static constexpr bool kEnabled = false;    // opt-in build (-DRM_PARKING): measured +12 % at the reference's pose (0,0,1),
#endif                                       // -8 % at (0,0,-3) and -12 % on the 512^2 config-4 frame
  const RenderArgs& a;
  int64_t li;
  RM_DEV bool operator()(int step, V3 p, bool unsettled) const {
    if (!a.park_ray) return false;
    const int k = step / park_stride(a.steps) - 2;
    if (k < 0 || k >= RM_PARK_LISTS) return false;
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int seg = k * RM_PARK_SHARDS + (int)(wave & (RM_PARK_SHARDS - 1));
    const unsigned long long m = __ballot(unsettled);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&a.minmax[RM_WORK_PARK_BASE + 32 * seg], (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const bool ok = unsettled && slot < (uint32_t)a.park_seg;     // segment full: the ray stays with its wave
    if (ok) {
      const int64_t at = (int64_t)seg * a.park_seg + slot;
      a.park_ray[at] = (int32_t)li;
      store3(a.park_p, at, p);
    }
    return ok;
  }
};

// kRecord = false: the plain inference frame (no trajectory, no per-ray outputs): the recording code and its live kernel
// arguments are compiled out of the march loop.  (Sharing one instantiation cost the headline frame 6 % in round 3:
// 208 -> 221 us on one box, 687 -> 997 v_readlane in the kernel from the extra SGPR pressure.)
template <class Cfg, bool kRecord = true>
__global__ void __launch_bounds__(256) k_render_fwd(RenderArgs a) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(a.scene, rm_smem, store);
  Tetra T = load_tetra(a.tetra);
  const bool early = a.flags & RM_FLAG_EARLY_OUT;
  MinMaxAcc mm{__builtin_inff(), -__builtin_inff(), false};
  const int64_t ntiles = wave_tiles(a);

  for (TileCursor tc = first_wave_tile(a, ntiles); tc.tile < ntiles; next_wave_tile(a, ntiles, tc)) {
    const int64_t tile = a.tile_order ? (int64_t)a.tile_order[tc.tile] : tc.tile;
    TileRays r = load_tile_rays(a, tile);
    int nexec;
    bool parked;
    // (a recorded trajectory is indexed by the wave tile's slot, not by the pixel: traj_store<true>)
    float* const traj = kRecord ? a.traj : nullptr;
    V3 p = march<decltype(scene), ParkToList, true>(scene, r.o, r.v, a.steps, early, traj, a.steps,
                                                    traj ? tile * 64 + (threadIdx.x & 63) : r.li, r.live, nexec,
                                                    ParkToList{a, r.li}, &parked);
    if (a.tile_cost && (threadIdx.x & 63) == 0) a.tile_cost[tile] = nexec;     // nexec is wave-uniform
    r.live = r.live && !parked;                 // a parked ray's pixel is written by k_render_parked
    finish_tile<decltype(scene), kRecord>(a, scene, T, r, p, nexec, mm);
  }
  if (a.minmax && (a.mode == RM_MODE_DISTANCE || a.mode == RM_MODE_PROXIMITY || a.mode == RM_MODE_LAPLACIAN))
    fold_minmax(a.minmax, mm.lo, mm.hi, mm.saw_nan);
}

// The rays parked by k_render_fwd, 64 of one list per wave: march the remaining steps (same deterministic
// iteration, its own cycle detection from there on), then distance / normals / shader / store like any other ray.
template <class Cfg>
__global__ void __launch_bounds__(256) k_render_parked(RenderArgs a) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(a.scene, rm_smem, store);
  Tetra T = load_tetra(a.tetra);
  const int W = a.cam.width, H = a.cam.height, rows = a.row_end - a.row_begin;
  const bool early = a.flags & RM_FLAG_EARLY_OUT;
  MinMaxAcc mm{__builtin_inff(), -__builtin_inff(), false};
  const int pstride = park_stride(a.steps);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  int64_t item = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);     // next wave item of this wave
  int64_t before = 0;                                                               // items of the segments already passed
  for (int seg = 0; seg < RM_PARK_LISTS * RM_PARK_SHARDS; ++seg) {
    const uint32_t c = a.minmax[RM_WORK_PARK_BASE + 32 * seg];
    const int count = (int)(c < (uint32_t)a.park_seg ? c : (uint32_t)a.park_seg);
    const int64_t n_items = (count + 63) >> 6;
    const int done = (seg / RM_PARK_SHARDS + 2) * pstride;                          // steps these rays have behind them
    for (; item < before + n_items; item += nwaves) {
      const int idx = (int)(item - before) * 64 + (threadIdx.x & 63);
      const bool live = idx < count;
      const int64_t at = (int64_t)seg * a.park_seg + (live ? idx : count - 1);
      TileRays r;
      r.live = live;
      r.li = a.park_ray[at];
      const int cam = (int)(r.li / ((int64_t)rows * W));
      const int64_t rem = r.li - (int64_t)cam * rows * W;
      const int row = (int)(rem / W), col = (int)(rem - (int64_t)row * W);
      const int64_t gi = ((int64_t)cam * H + (row + a.row_begin)) * W + col;
      V3 o, v;
      if (a.cam.dtype == RM_DTYPE_F16) load_ray_and_pose<_Float16>(a, gi, cam, o, v, r.ps);
      else load_ray_and_pose<float>(a, gi, cam, o, v, r.ps);
      r.o = qrot(o, r.ps.w, r.ps.qv) + r.ps.t;
      r.v = qrot(v, r.ps.w, r.ps.qv);
      int nexec;
      V3 p = march(scene, load3(a.park_p, at), r.v, a.steps - done, early, nullptr, 0, r.li, live, nexec);
      finish_tile(a, scene, T, r, p, done + nexec, mm);
    }
    before += n_items;
  }
  if (a.minmax && (a.mode == RM_MODE_DISTANCE || a.mode == RM_MODE_PROXIMITY || a.mode == RM_MODE_LAPLACIAN))
    fold_minmax(a.minmax, mm.lo, mm.hi, mm.saw_nan);
}

// ---- ray regeneration (RM_FLAG_REGEN) -----------------------------------------------------------------------
// The wave-uniform exit of `march` keeps 64 lanes busy until the LAST ray of the tile has a proven period: at the
// reference's default pose (0,0,1) 2.49 ray-steps are executed per ray-step needed (profiles/settle_probe.py).  Here a
// wave is a pool of 64 ray slots instead of a tile: a lane whose ray has reached its final iterate stores it and is
// handed the next ray of a queue (one returning atomic per wave and refill, for all free lanes together), so the
// lanes stay full while any ray is left.  Only the march lives here: the final iterates go to `p_final` and
// k_render_finish does distance / normals / shader per 8x8 tile with full, coherent waves (those ~7 uncullable
// evaluations would otherwise run at the pool's partial occupancy).
//
// Per-lane bookkeeping replaces the wave-uniform one: local step count k (a multiple of 4 at every look), snapshot
// for Brent's cycle search refreshed at k = 4, 8, 16, ..., and `stop` = the local step at which the lane holds
// exactly the iterate the full S-step loop would end on (k + (S - k) mod lambda once a period lambda is proven at k).
// The iteration itself is the same instruction stream per ray, so the image is bit-identical to k_render_fwd's.
#ifndef RM_REGEN_MIN_FREE
#define RM_REGEN_MIN_FREE 16      // refill when at least this many lanes of the wave are idle
#endif

// Queue q owns the dealing positions q, q + 64, q + 128, ... like the tile queues (so every queue hands its tiles out
// in the global dealing order: longest first when a tile_order is given), 64 ray slots per position.  A draw returns
// queue-local slots; slot s of queue q is lane s % 64 of position q + 64 (s / 64).
RM_DEV uint32_t slot_queue_size(int64_t ntiles, int q) { return 64u * queue_size(ntiles, q); }

// `m` consecutive slots of one queue for this wave (fewer at its end: `count`); -1 when every queue is drained.
// Same structure as grab_wave_tile: counters only grow, every slot is handed out once, the grid always drains.
RM_DEV int64_t grab_ray_slots(uint32_t* work, int64_t ntiles, int m, int& q, int& count) {
  const int lane = threadIdx.x & 63;
  uint32_t* ctr = work + RM_WORK_QUEUE_BASE;
  for (int attempt = 0; attempt < 4 * RM_WORK_QUEUES; ++attempt) {
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&ctr[q * RM_WORK_QUEUE_STRIDE], (uint32_t)m);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) + 64u * first_positions(q);
    const uint32_t size = slot_queue_size(ntiles, q);
    if (t < size) {
      count = (size - t < (uint32_t)m) ? (int)(size - t) : m;
      return (int64_t)t;
    }
    uint32_t c = __hip_atomic_load(&ctr[lane * RM_WORK_QUEUE_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned long long avail = __ballot(c + 64u * first_positions(lane) < slot_queue_size(ntiles, lane));
    if (avail == 0ull) return -1;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int r = (int)(((wave * 2654435761u) >> 20) + attempt * 17) & 63;
    unsigned long long rot = (r == 0) ? avail : ((avail >> r) | (avail << (64 - r)));
    q = (r + __builtin_ctzll(rot)) & 63;
  }
  for (int k = 0; k < RM_WORK_QUEUES; ++k) {     // not reached in practice: definitive sweep
    uint32_t t = 0;
    if (lane == 0) t = atomicAdd(&ctr[k * RM_WORK_QUEUE_STRIDE], (uint32_t)m);
    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t) + 64u * first_positions(k);
    const uint32_t size = slot_queue_size(ntiles, k);
    if (t < size) {
      q = k;
      count = (size - t < (uint32_t)m) ? (int)(size - t) : m;
      return (int64_t)t;
    }
  }
  return -1;
}

template <class Cfg>
__global__ void __launch_bounds__(256) k_march_regen(RenderArgs a) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(a.scene, rm_smem, store);
  const int W = a.cam.width, H = a.cam.height, rows = a.row_end - a.row_begin;
  const int S = a.steps;                          // a multiple of 4 (checked on the host)
  const int lane = threadIdx.x & 63;
  const int64_t ntiles = wave_tiles(a);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  int q = (int)(wave & (RM_WORK_QUEUES - 1));
  bool open = true, first = true;
  // the ray of this lane
  bool has = false;
  int64_t li = 0;
  int item = 0;                                   // ray slot: tile * 64 + lane of the tile
  V3 p = mk3(0.0f, 0.0f, 0.0f), v = p, snap = p;
  int k = 0, stop = 0, lambda = 0, snap_step = 0, next_snap = 4;
  float move = __builtin_nanf(""), vn = 0.0f;
  int it = 0;                                     // steps this wave has walked (cull knowledge is dropped every 16th)
#ifdef RM_REGEN_STATS
  unsigned st_groups = 0, st_lanes = 0, st_refills = 0, st_tail_groups = 0, st_tail_lanes = 0;
#endif
  for (;;) {
    if (has && k >= stop) {                       // final iterate reached: hand it to k_render_finish
      store3(a.p_final, li, p);
      if (a.tile_cost) a.tile_cost[item] = stop;     // steps this ray needed (slots outside the image: zeroed by the host)
      has = false;
    }
    const unsigned long long freem = __ballot(!has);
    const int nfree = __popcll(freem);
    if (!open && nfree == 64) break;
#ifdef RM_REGEN_STATS
    if (open && nfree >= RM_REGEN_MIN_FREE) ++st_refills;
    else { ++st_groups; st_lanes += 64 - nfree; if (!open) { ++st_tail_groups; st_tail_lanes += 64 - nfree; } }
#endif
    if (open && nfree >= RM_REGEN_MIN_FREE) {
      int count = 0;
      int64_t base = -1;
      if (first) {                                // static first draw: position wave/64 of the home queue, no atomic
        first = false;
        const uint32_t t = (uint32_t)(wave >> 6) * 64u, size = slot_queue_size(ntiles, q);
        if (t < size) { base = (int64_t)t; count = (size - t < 64u) ? (int)(size - t) : 64; }
      }
      if (base < 0) base = grab_ray_slots(a.minmax, ntiles, nfree, q, count);
      if (base < 0) {
        open = false;
      } else {
        const int rank = __popcll(freem & ((1ull << lane) - 1ull));
        const int64_t slot = base + rank;
        int cam = 0, row = 0, col = 0;
        bool take = !has && rank < count;
        // dealing order (tile_order; with RM_FLAG_ORDER_PER_RAY one entry per RAY SLOT = tile * 64 + lane of the
        // tile): the rays that march longest first, so that what is left when the queues run dry are rays that settle
        // within a few steps -- a pool cannot refill its idle lanes any more by then
        const int64_t tpos = (int64_t)q + (slot >> 6) * RM_WORK_QUEUES;        // dealing position of the tile
        const int64_t pos = tpos * 64 + (slot & 63);                            // ... of the ray slot
        if (take) {
          if (!a.tile_order) item = (int)pos;
          else if (a.flags & RM_FLAG_ORDER_PER_RAY) item = a.tile_order[pos];
          else item = a.tile_order[tpos] * 64 + (int)(slot & 63);
        }
        take = take && ray_of_tile_lane(a, item >> 6, item & 63, cam, row, col);
        if (take) {
          li = ((int64_t)cam * rows + row) * W + col;
          const int64_t gi = ((int64_t)cam * H + (row + a.row_begin)) * W + col;
          V3 o, d;
          Pose ps;
          if (a.cam.dtype == RM_DTYPE_F16) load_ray_and_pose<_Float16>(a, gi, cam, o, d, ps);
          else load_ray_and_pose<float>(a, gi, cam, o, d, ps);
          p = qrot(o, ps.w, ps.qv) + ps.t;        // PinholeCamera.forward (ray_marching.py:58-62)
          v = qrot(d, ps.w, ps.qv);
          vn = 1.0001f * __builtin_amdgcn_sqrtf(__builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
          move = __builtin_nanf("");
          snap = p; snap_step = 0; next_snap = 4; lambda = 0; k = 0; stop = S;
          has = true;
        }
      }
      continue;                                   // (a ray of zero steps retires at once; an empty draw tries again)
    }
    const bool act = has && k < stop;
#if RM_PRIO_REGEN > 0
    // rays that have marched long are the critical path of the launch (nothing can refill their lanes once the queues
    // are dry): pools holding one are issued first (32-primitive 8K band 11.08 -> 10.42 ms; 1080p scene 2 unchanged)
    if (__any(act && k >= RM_PRIO_REGEN)) __builtin_amdgcn_s_setprio(3);
    else __builtin_amdgcn_s_setprio(0);
#endif
    if constexpr (!decltype(scene.prog)::kNeedsFullWave) {
      if (act) {
        V3 prev = p;
        for (int j = 0; j < 4; ++j) {
          const float f = scene.eval_near(p, ((it + j) & 15) ? move : __builtin_nanf(""));
          move = __builtin_fmaf(fabsf(f), vn, 4e-6f);
          prev = p;
          p = step_point(p, v, f);
        }
        k += 4;
        if (lambda == 0) {
          const int found = same_bits(p, prev) ? 1 : (same_bits(p, snap) ? (k - snap_step) : 0);
          if (found) { lambda = found; stop = k + (S - k) % found; }
        }
        if (k == next_snap) { snap = p; snap_step = k; next_snap <<= 1; }
      }
    } else {
      // Programs with CULL_LSE evaluate the scene with EVERY lane of the wave active (the wave-wide reductions of
      // lse_cull_mask need that): a lane without a marching ray shadows the first active lane's ray -- the same point,
      // hence the same votes in every wave-uniform cull test and no influence on the bounds -- and throws the result
      // away.  (Costs the pools ~5 % -- 0.311 -> 0.327 ms at (0,0,1) when every program paid it -- so only those do.)
      const unsigned long long actm = __ballot(act);
      if (actm) {
        const int src = __builtin_ctzll(actm);
        auto from = [&](float x) {
          return act ? x : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), src));
        };
        V3 pe = mk3(from(p.x), from(p.y), from(p.z));
        const V3 ve = mk3(from(v.x), from(v.y), from(v.z));
        const float vne = from(vn);
        float movee = from(move);
        V3 prev = pe;
        for (int j = 0; j < 4; ++j) {
          const float f = scene.eval_near(pe, ((it + j) & 15) ? movee : __builtin_nanf(""));
          movee = __builtin_fmaf(fabsf(f), vne, 4e-6f);
          prev = pe;
          pe = step_point(pe, ve, f);
        }
        if (act) {
          p = pe; move = movee;
          k += 4;
          if (lambda == 0) {
            const int found = same_bits(p, prev) ? 1 : (same_bits(p, snap) ? (k - snap_step) : 0);
            if (found) { lambda = found; stop = k + (S - k) % found; }
          }
          if (k == next_snap) { snap = p; snap_step = k; next_snap <<= 1; }
        } else {
          move = __builtin_nanf("");      // the tracked cull bounds now describe the shadowed ray, not this lane's next one
        }
      }
    }
    it += 4;
  }
#ifdef RM_REGEN_STATS
  if (lane == 0) {
    atomicAdd(&a.minmax[8], st_groups); atomicAdd(&a.minmax[9], st_lanes); atomicAdd(&a.minmax[10], st_refills);
    atomicAdd(&a.minmax[11], st_tail_groups); atomicAdd(&a.minmax[12], st_tail_lanes);
    atomicMax(&a.minmax[13], st_groups); atomicMax(&a.minmax[14], st_tail_groups);
  }
#endif
}

// distance / normals / shader of the final iterates k_march_regen left in p_final: one 8x8 tile per wave, static
// striding (every tile costs the same here)
template <class Cfg>
__global__ void __launch_bounds__(256) k_render_finish(RenderArgs a) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(a.scene, rm_smem, store);
  Tetra T = load_tetra(a.tetra);
  MinMaxAcc mm{__builtin_inff(), -__builtin_inff(), false};
  const int64_t ntiles = wave_tiles(a);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < ntiles; tile += nwaves) {
    TileRays r = load_tile_rays(a, tile);
    const V3 p = load3(a.p_final, r.li);
    finish_tile(a, scene, T, r, p, a.steps, mm, false);
  }
  if (a.minmax && (a.mode == RM_MODE_DISTANCE || a.mode == RM_MODE_PROXIMITY || a.mode == RM_MODE_LAPLACIAN))
    fold_minmax(a.minmax, mm.lo, mm.hi, mm.saw_nan);
}

// Dealing score of a tile from the per-ray step counts k_march_regen recorded (cost[tile * 64 + lane]): which rays
// march long is noise from one pose to the next (whether an iterate falls into a short exact cycle), how MANY of a
// tile's rays do is not (correlation 0.96 - 0.999 across a one-pixel camera move, profiles/regen_costmap.py).
// Tiles with long rays (>= 3/4 of the steps) come first, most of them first -- classes 31 .. 17; then, class 16,
// the tiles WITHOUT long rays within `reach` tiles of one: when the camera moves by up to 8 * reach pixels until the
// order is renewed these are where long rays turn up, and a long ray dealt at the very end of the order is the
// worst case (measured: 2 % of such tiles cost 13 % of the frame); then the rest by the longest ray in their
// neighbourhood -- classes 15 .. 0.  Pass 1: one wave per tile, raw = long rays << 16 | longest ray.
__global__ void __launch_bounds__(256) k_tile_score_raw(const int32_t* __restrict__ ray_cost, int64_t n_tiles, int max_cost,
                                                        int32_t* __restrict__ raw) {
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t tile = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); tile < n_tiles; tile += nwaves) {
    const int c = ray_cost[tile * 64 + lane];
    const int n_long = __popcll(__ballot(4 * (long long)c >= 3 * (long long)max_cost && c > 0));
    int mx = c;
    for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(mx, o, 64); mx = other > mx ? other : mx; }
    mx = mx < 0 ? 0 : (mx > max_cost ? max_cost : mx);
    if (mx > 0xffff) mx = 0xffff;
    if (lane == 0) raw[tile] = (n_long << 16) | mx;
  }
}

// Pass 2: classes from the raw values of the (2 reach + 1)^2 neighbourhood inside the tile grid of the camera
__global__ void __launch_bounds__(256) k_tile_score_classes(const int32_t* __restrict__ raw, int64_t n_tiles, int tiles_x,
                                                            int tiles_y, int reach, int max_cost, int32_t* __restrict__ score) {
  const int64_t per_cam = (int64_t)tiles_x * tiles_y;
  for (int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tile < n_tiles; tile += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cam0 = (tile / per_cam) * per_cam;
    const int ty = (int)((tile - cam0) / tiles_x), tx = (int)((tile - cam0) - (int64_t)ty * tiles_x);
    const int own = raw[tile];
    int near_long = 0, near_max = own & 0xffff;
    for (int dy = -reach; dy <= reach; ++dy)
      for (int dx = -reach; dx <= reach; ++dx) {
        const int y = ty + dy, x = tx + dx;
        if (y < 0 || y >= tiles_y || x < 0 || x >= tiles_x) continue;
        const int r = raw[cam0 + (int64_t)y * tiles_x + x];
        near_long |= r >> 16;
        near_max = (r & 0xffff) > near_max ? (r & 0xffff) : near_max;
      }
    const int n_long = own >> 16;
    const int mx = near_max > max_cost ? max_cost : near_max;
    score[tile] = n_long ? 17 + ((n_long - 1) * 15) / 64 : (near_long ? 16 : (int)(((long long)mx * 16) / (max_cost + 1)));
  }
}

// Longest-first dealing order from the previous frame's step counts: STABLE counting sort by descending cost class
// (32 classes), so items of one class keep their natural, spatially coherent order (a scatter through atomics shuffled
// the tiles and cost the headline frame 5 %).  Any permutation renders the same image.
//
// Block b owns a contiguous chunk of the items, thread t a contiguous run of it, with its own 32 counters in LDS
// (bin-major [32][1024]).  k_order_count leaves the per-(block, class) totals in `counts`; k_order_scatter recounts,
// scans every class row over the threads (one wave per row, 16 rows at a time) and adds the start of its
// (class, block) cell in the global class-major, block-minor layout.  One block needs no `counts`.
RM_DEV int order_bin(int c, int max_cost) {
  c = c < 0 ? 0 : (c > max_cost ? max_cost : c);
  return 31 - (int)(((long long)c * 32) / (max_cost + 1));                  // bin 0 = the most expensive class
}

__global__ void __launch_bounds__(1024) k_order_count(const int32_t* __restrict__ cost, int64_t n, int max_cost,
                                                      int32_t* __restrict__ counts) {
  __shared__ int s_tot[32];
  const int t = threadIdx.x;
  const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = (int64_t)blockIdx.x * chunk, b1 = (b0 + chunk < n) ? b0 + chunk : n;
  if (t < 32) s_tot[t] = 0;
  __syncthreads();
  int mine[32];
#pragma unroll
  for (int b = 0; b < 32; ++b) mine[b] = 0;
  // strided over the block here (coalesced): only totals are needed
  for (int64_t i = b0 + t; i < b1; i += blockDim.x) {
    const int bin = order_bin(cost[i], max_cost);
#pragma unroll
    for (int b = 0; b < 32; ++b) mine[b] += (bin == b);
  }
#pragma unroll
  for (int b = 0; b < 32; ++b) {
    int v = mine[b];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((t & 63) == 0 && v) atomicAdd(&s_tot[b], v);
  }
  __syncthreads();
  if (t < 32) counts[blockIdx.x * 32 + t] = s_tot[t];
}

// `counts` may be null when the grid is one block
__global__ void __launch_bounds__(1024) k_order_scatter(const int32_t* __restrict__ cost, int64_t n, int max_cost,
                                                        const int32_t* __restrict__ counts, int32_t* __restrict__ order) {
  constexpr int kBins = 32;
  extern __shared__ int s_hist[];                         // [kBins][blockDim.x], then [kBins] row totals, [kBins] cell starts
  const int nt = blockDim.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = nt >> 6;
  int* s_tot = s_hist + kBins * nt;
  int* s_base = s_tot + kBins;
  const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = (int64_t)blockIdx.x * chunk, b1 = (b0 + chunk < n) ? b0 + chunk : n;
  const int64_t run = (chunk + nt - 1) / nt;
  const int64_t lo = (b0 + t * run < b1) ? b0 + t * run : b1, hi = (lo + run < b1) ? lo + run : b1;
  for (int b = 0; b < kBins; ++b) s_hist[b * nt + t] = 0;
  for (int64_t i = lo; i < hi; ++i) s_hist[order_bin(cost[i], max_cost) * nt + t] += 1;
  __syncthreads();
  // exclusive scan of every class row over the threads: 64 entries at a time (conflict-free), carry in between
  for (int b = wave; b < kBins; b += nwaves) {
    int* row = s_hist + b * nt;
    int carry = 0;
    for (int k = 0; k < nt; k += 64) {
      const int v = row[k + lane];
      int incl = v;
      for (int o = 1; o < 64; o <<= 1) {
        const int up = __shfl_up(incl, o, 64);
        if (lane >= o) incl += up;
      }
      row[k + lane] = carry + incl - v;
      carry += __shfl(incl, 63, 64);
    }
    if (lane == 0) s_tot[b] = carry;
  }
  __syncthreads();
  if (t < 64) {
    // start of cell (class t, this block): all items of more expensive classes, then this class in earlier blocks
    int total = 0, before = 0;
    if (t < kBins) {
      if (gridDim.x == 1) total = s_tot[t];
      else
        for (int b = 0; b < (int)gridDim.x; ++b) {
          const int c = counts[b * 32 + t];
          total += c;
          if (b < (int)blockIdx.x) before += c;
        }
    }
    int incl = total;
    for (int o = 1; o < 64; o <<= 1) {
      const int up = __shfl_up(incl, o, 64);
      if (t >= o) incl += up;
    }
    if (t < kBins) s_base[t] = (incl - total) + before;
  }
  __syncthreads();
  for (int64_t i = lo; i < hi; ++i) {
    const int b = order_bin(cost[i], max_cost);
    order[s_base[b] + s_hist[b * nt + t]++] = (int32_t)i;
  }
}

// second pass for the globally normalised shaders
__global__ void k_shade_finish(const float* src, void* image, int dt, int64_t n, const uint32_t* __restrict__ minmax, int mode,
                               int round_dt) {
  float lo, hi;
  load_minmax(minmax, lo, hi);
  const float gamma = (float)(1.0 / 2.33);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float x = src[3 * i];
    float y;
    if (mode == RM_MODE_LAPLACIAN) {   // shader.py:83-88
      y = rm_pow(t_clamp((((x / hi) * -1.0f) + 1.0f) / 2.0f, 0.0f, 1.0f), gamma);
    } else {                           // shader.py:34-38 / 51-55
      y = rm_pow((x - lo) / (hi - lo), gamma);
    }
    if (dt == RM_DTYPE_RGBA_F32) store_rgba(image, i, mk3(y, y, y), round_dt);
    else store3_t(image, i, mk3(y, y, y), dt);
  }
}

// ---- VJP of the globally normalised shaders' normalisation (shader.py:33-38, 51-55, 81-89) ------------------------------
// The normalisation is a reduction over every pixel; its VJP, in the order autograd walks it, needs two sums and two
// counts over the frame:
//   distance / proximity:  y = ((x - lo) / (hi - lo))^gamma with lo = min x, hi = max x (both x.min() nodes of the reference):
//       gx = gy gamma (a / r)^(gamma - 1),  ga = gx / r,  g_r = sum(-gx a / r^2),  g_lo = -sum(ga) - g_r,
//       dL/dx = ga + [x == hi] g_r / #[x == hi] + [x == lo] g_lo / #[x == lo]                 (a = x - lo, r = hi - lo)
//   laplacian:  y = clamp((x / hi * -1 + 1) / 2, 0, 1)^gamma with hi = max |x|:
//       gu = gy gamma clamp(u)^(gamma - 1) [0 <= u <= 1],  ga = gu / 2 * -1,  g_hi = sum(ga * (-x / hi^2)),
//       dL/dx = ga / hi + [|x| == hi] (g_hi / #[|x| == hi]) sign(x)
// x^(gamma - 1) is infinite at x = 0: the inf - inf and inf * 0 the reference's gradient carries at the rays of the
// minimum / maximum are reproduced by doing the same IEEE operations (products with masks are products, not selects).
// Pass A: per-pixel terms, ga into out[..., 0], block sums of the four quantities in a fixed order -> partials[block][4].
// Pass B: every block adds the partials up in block order (deterministic), then the elementwise combination.
struct NormBwdTerms {
  float ga, t1;
  bool top, bottom;
};
RM_DEV NormBwdTerms norm_bwd_terms(int mode, float x, V3 g, float lo, float hi) {
  const float gamma = (float)(1.0 / 2.33), gm1 = (float)(1.0 / 2.33 - 1.0);
  const float gy = (g.x + g.y) + g.z;                    // grad of expand(-1, H, W, 3): the channels add up
  NormBwdTerms o;
  if (mode == RM_MODE_LAPLACIAN) {
    const float u = (((x / hi) * -1.0f) + 1.0f) / 2.0f;
    const float xc = t_clamp(u, 0.0f, 1.0f);
    const float mask = (u >= 0.0f && u <= 1.0f) ? 1.0f : 0.0f;
    const float gu = (gy * (gamma * rm_pow(xc, gm1))) * mask;
    const float ga = (gu * 0.5f) * -1.0f;
    o.ga = ga / hi;
    o.t1 = ga * (-x / (hi * hi));
    o.top = fabsf(x) == hi;
    o.bottom = false;
  } else {
    const float r = hi - lo, a = x - lo;
    const float gx = gy * (gamma * rm_pow(a / r, gm1));
    o.ga = gx / r;
    o.t1 = (-gx * a) / (r * r);
    o.top = x == hi;
    o.bottom = x == lo;
  }
  return o;
}

RM_DEV float block_sum_256(float v, float* red) {      // fixed tree over the 256 threads of a block
  __syncthreads();
  red[threadIdx.x] = v;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  return red[0];
}

__global__ void __launch_bounds__(256) k_shade_norm_bwd_a(const float* __restrict__ raw, const float* __restrict__ grad_image,
                                                          const float* __restrict__ lohi, int mode, float* __restrict__ out,
                                                          float* __restrict__ partials, int64_t n) {
  __shared__ float red[256];
  const float lo = lohi[0], hi = lohi[1];
  float s_t1 = 0.0f, s_ga = 0.0f, c_top = 0.0f, c_bot = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const NormBwdTerms t = norm_bwd_terms(mode, raw[3 * i], load3(grad_image, i), lo, hi);
    out[3 * i] = t.ga;
    s_t1 += t.t1; s_ga += t.ga;
    c_top += t.top ? 1.0f : 0.0f; c_bot += t.bottom ? 1.0f : 0.0f;      // exact below 2^24 pixels per thread
  }
  const float a = block_sum_256(s_t1, red), b = block_sum_256(s_ga, red), c = block_sum_256(c_top, red), d = block_sum_256(c_bot, red);
  if (threadIdx.x == 0) {
    float* dst = partials + 4 * blockIdx.x;
    dst[0] = a; dst[1] = b; dst[2] = c; dst[3] = d;
  }
}

__global__ void __launch_bounds__(256) k_shade_norm_bwd_b(const float* __restrict__ raw, const float* __restrict__ lohi, int mode,
                                                          float* __restrict__ out, const float* __restrict__ partials,
                                                          int n_partials, int64_t n) {
  // Every block adds up the block sums of the first launch itself, in ONE fixed order (thread t takes rows t, t + 256,
  // ..., then the tree of block_sum_256): counts are sums of integers held in floats (exact), the two float sums come
  // out the same in every block.  (Four threads walking 1024 rows one after the other made this launch 95 us.)
  __shared__ float red[256];
  float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (int b = threadIdx.x; b < n_partials; b += 256) {
    const float4 r = *reinterpret_cast<const float4*>(partials + 4 * b);
    acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w;
  }
  const float g_r = block_sum_256(acc.x, red), s_ga = block_sum_256(acc.y, red);
  const float n_top = block_sum_256(acc.z, red), n_bot = block_sum_256(acc.w, red);
  const float lo = lohi[0], hi = lohi[1];
  const float g_lo = -s_ga - g_r;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x = raw[3 * i];
    float v = out[3 * i];
    if (mode == RM_MODE_LAPLACIAN) {
      v = v + ((fabsf(x) == hi) ? (g_r / n_top) * sgn0(x) : 0.0f);
    } else {
      v = (v + ((x == hi) ? g_r / n_top : 0.0f)) + ((x == lo) ? g_lo / n_bot : 0.0f);
    }
    out[3 * i] = v; out[3 * i + 1] = 0.0f; out[3 * i + 2] = 0.0f;
  }
}

// workspace: words 0-2 global min / max / NaN flag, the rest zero (tile queue counters)
__global__ void k_minmax_init(uint32_t* mm) {
  mm += (size_t)blockIdx.x * RM_WORK_WORDS;       // one block per workspace (rm_minmax_init_many)
  for (int i = threadIdx.x; i < RM_WORK_WORDS; i += blockDim.x) mm[i] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) { mm[0] = f2ord(__builtin_inff()); mm[1] = f2ord(-__builtin_inff()); }
  if (threadIdx.x < RM_WORK_MM_SLOTS) {
    mm[RM_WORK_MM_BASE + 32 * threadIdx.x] = f2ord(__builtin_inff());
    mm[RM_WORK_MM_BASE + 32 * threadIdx.x + 1] = f2ord(-__builtin_inff());
  }
}
// one wave
__global__ void k_minmax_decode(const uint32_t* mm, float* lohi) {
  float lo, hi;
  load_minmax(mm, lo, hi);
  if (threadIdx.x == 0) { lohi[0] = lo; lohi[1] = hi; }
}
// one wave: the given pair becomes the global value (words 0-2; the partial triples go back to neutral)
__global__ void k_minmax_encode(const float* lohi, uint32_t* mm) {
  if (threadIdx.x == 0) {
    bool isn = (lohi[0] != lohi[0]) || (lohi[1] != lohi[1]);
    mm[0] = f2ord(lohi[0]); mm[1] = f2ord(lohi[1]); mm[2] = isn ? 1u : 0u; mm[3] = 0u;
  }
  if (threadIdx.x < RM_WORK_MM_SLOTS) {
    mm[RM_WORK_MM_BASE + 32 * threadIdx.x] = f2ord(__builtin_inff());
    mm[RM_WORK_MM_BASE + 32 * threadIdx.x + 1] = f2ord(-__builtin_inff());
    mm[RM_WORK_MM_BASE + 32 * threadIdx.x + 2] = 0u;
  }
}

// standalone PinholeCamera.forward
__global__ void k_camera_fwd(RmCamera cam, const void* __restrict__ orientation, const void* __restrict__ translation,
                             void* __restrict__ out_pos, void* __restrict__ out_dirs, void* __restrict__ frames) {
  const int dt = cam.dtype;
  int64_t per_cam = (int64_t)cam.height * cam.width;
  int64_t n = per_cam * cam.num_cameras;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i / per_cam);
    Pose ps = load_pose(orientation, translation, c, dt);
    store3_t(out_pos, i, qrot(load3_t(cam.ray_positions, i, dt), ps.w, ps.qv) + ps.t, dt);
    store3_t(out_dirs, i, qrot(load3_t(cam.ray_directions, i, dt), ps.w, ps.qv), dt);
  }
  // QuaternionToSO3 (quaternion.py:114-124)
  if (frames && blockIdx.x == 0 && (int)threadIdx.x < cam.num_cameras) {
    int c = threadIdx.x;
    Pose ps = load_pose(orientation, translation, c, dt);
    float w = ps.w, x = ps.qv.x, y = ps.qv.y, z = ps.qv.z;
    float ww = w * w, wx = w * x, wy = w * y, wz = w * z, xx = x * x, xy = x * y, xz = x * z, yy = y * y, yz = y * z, zz = z * z;
    const float f[9] = {((ww + xx) - yy) - zz, 2.0f * (xy - wz), 2.0f * (wy + xz),
                        2.0f * (xy + wz), ((ww - xx) + yy) - zz, 2.0f * (yz - wx),
                        2.0f * (xz - wy), 2.0f * (wx + yz), ((ww - xx) - yy) + zz};
    for (int k = 0; k < 9; ++k) st_t(frames, 9 * c + k, f[k], dt);
  }
}

// VJP of the per-pixel shaders that have one: Lambertian (mode 0), vignette (3), normal (4).
// grad_image is [n, C] with C = 1 (modes 0, 3) or 3 (mode 4); outputs may be null.
struct ShadeBwdArgs {
  const float *dirs, *normals, *frames, *grad_image;
  float *grad_dirs, *grad_normals;
  int32_t mode;
  int64_t n, per_camera;
};

__global__ void k_shade_bwd(ShadeBwdArgs a) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (int64_t)gridDim.x * blockDim.x) {
    V3 gv = mk3(0.0f, 0.0f, 0.0f), gn = mk3(0.0f, 0.0f, 0.0f);
    if (a.mode == RM_MODE_LAMBERTIAN) {        // clamp(-(v.n), 0, 1): grad passes on the closed interval
      V3 v = load3(a.dirs, i), n = load3(a.normals, i);
      float c = -dot_seq(v, n);
      float g = (c >= 0.0f && c <= 1.0f) ? a.grad_image[i] : 0.0f;
      gv = mk3(-g * n.x, -g * n.y, -g * n.z);
      gn = mk3(-g * v.x, -g * v.y, -g * v.z);
    } else if (a.mode == RM_MODE_VIGNETTE) {   // (v . col2)^3
      int cam = (int)(i / a.per_camera);
      V3 v = load3(a.dirs, i);
      V3 c2 = mk3(a.frames[9 * cam + 2], a.frames[9 * cam + 5], a.frames[9 * cam + 8]);
      float d = dot_seq(v, c2);
      float g = a.grad_image[i] * (3.0f * (d * d));
      gv = mk3(g * c2.x, g * c2.y, g * c2.z);
    } else {                                   // clamp(|n|, 0, 1) per channel
      V3 n = load3(a.normals, i), g = load3(a.grad_image, i);
      gn = mk3((fabsf(n.x) <= 1.0f) ? g.x * sgn0(n.x) : 0.0f, (fabsf(n.y) <= 1.0f) ? g.y * sgn0(n.y) : 0.0f,
               (fabsf(n.z) <= 1.0f) ? g.z * sgn0(n.z) : 0.0f);
    }
    if (a.grad_dirs) store3(a.grad_dirs, i, gv);
    if (a.grad_normals) store3(a.grad_normals, i, gn);
  }
}

// VJP of PinholeCamera.forward w.r.t. the pose.  pos = rot(o_c, q) + t, dir = rot(v_c, q) with
// rot(V, q) = V + w T + u x T, T = 2 u x V.  Per ray: g_t = g_pos; g_w = g.T; g_T = w g + g x u;
// g_u = T x g + 2 V x g_T (summed over the origin and direction terms).  Rays of rows
// [row_begin,row_end); block partial sums [camera][block][7] in a fixed order (deterministic).
__global__ void k_camera_bwd(RmCamera cam, const float* __restrict__ orientation, const float* __restrict__ gpos,
                             const float* __restrict__ gdirs, float* __restrict__ partials, int row_begin,
                             int row_end, int blocks_per_cam) {
  __shared__ float red[7][256];
  const int c = blockIdx.x / blocks_per_cam, b = blockIdx.x % blocks_per_cam;
  const int W = cam.width, rows = row_end - row_begin;
  const int64_t per_cam = (int64_t)rows * W;
  float w = orientation[4 * c];
  V3 u = mk3(orientation[4 * c + 1], orientation[4 * c + 2], orientation[4 * c + 3]);
  float acc[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int64_t r = (int64_t)b * blockDim.x + threadIdx.x; r < per_cam; r += (int64_t)blocks_per_cam * blockDim.x) {
    int row = (int)(r / W), col = (int)(r - (int64_t)row * W);
    int64_t li = (int64_t)c * per_cam + r;
    int64_t gi = ((int64_t)c * cam.height + (row + row_begin)) * W + col;
    for (int k = 0; k < 2; ++k) {
      const float* gsrc = k ? gdirs : gpos;
      if (!gsrc) continue;
      V3 g = load3(gsrc, li);
      V3 V = load3(static_cast<const float*>(k ? cam.ray_directions : cam.ray_positions), gi);
      V3 T = 2.0f * cross(u, V);
      V3 gT = w * g + cross(g, u);
      V3 gu = cross(T, g) + 2.0f * cross(V, gT);
      acc[0] += (g.x * T.x + g.y * T.y) + g.z * T.z;
      acc[1] += gu.x; acc[2] += gu.y; acc[3] += gu.z;
      if (k == 0) { acc[4] += g.x; acc[5] += g.y; acc[6] += g.z; }
    }
  }
  for (int k = 0; k < 7; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2)
      for (int k = 0; k < 7; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x < 7) partials[((int64_t)c * blocks_per_cam + b) * 7 + threadIdx.x] = red[threadIdx.x][0];
}

__global__ void __launch_bounds__(256) k_camera_bwd_finish(const float* __restrict__ partials, int blocks_per_cam, int num_cameras,
                                    float* __restrict__ gq, float* __restrict__ gt) {
  // one block of 256 threads per camera: thread t takes the rows t, t + 256, ... of that camera's block sums, then a
  // fixed tree (seven threads walking the rows one after the other were a chain of dependent loads)
  __shared__ float red[7][256];
  const int c = blockIdx.x;
  if (c >= num_cameras) return;
  float acc[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
  for (int b = threadIdx.x; b < blocks_per_cam; b += 256) {
    const float* row = partials + ((int64_t)c * blocks_per_cam + b) * 7;
    for (int k = 0; k < 7; ++k) acc[k] += row[k];
  }
  for (int k = 0; k < 7; ++k) red[k][threadIdx.x] = acc[k];
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2)
      for (int k = 0; k < 7; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s2];
    __syncthreads();
  }
  const int k = threadIdx.x;
  if (k < 4) { if (gq) gq[4 * c + k] = red[k][0]; }
  else if (k < 7 && gt) gt[3 * c + (k - 4)] = red[k][0];
}

// appends the active lanes of a wave to the deferred-ray list (one returning atomic per wave)
struct DeferToList {
  const RenderArgs& a;
  int64_t li, tslot;
  RM_DEV bool operator()(int i, V3 lam, V3 gv, bool active) const {
    if (a.hard_cap <= 0 || !a.minmax) return false;
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(active);
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&a.minmax[RM_WORK_HARD_COUNT], (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    const bool ok = active && slot < (uint32_t)a.hard_cap;       // list full: the wave walks these rays itself
    if (ok) {
      a.hard_ray[slot] = (int32_t)li;
      a.hard_slot[slot] = (int32_t)tslot;
      a.hard_step[slot] = i;
      float* st = a.hard_state + 8 * (int64_t)slot;
      st[0] = lam.x; st[1] = lam.y; st[2] = lam.z; st[3] = gv.x; st[4] = gv.y; st[5] = gv.z;
    }
    return ok;
  }
};

// VJP of the fused frame w.r.t. scene parameters (modes 0 and 4).
// kKind 0: the per-pixel shaders.  The globally normalised ones (modes 1, 2, 5: shader.py:27-38, 45-55, 81-89) have a
// normalisation that is a reduction over every pixel -- and every rank of a row-tiled render; the host differentiates
// it (ray_marching_amd/ops.py) and grad_image[..., 0] then is dL/d(un-normalised value) of the ray:
//   kKind 3, mode 1: log(clamp(|origin - p|, 1e-2, inf)) -- no scene evaluation, but three more live registers;
//   kKind 2, mode 2: log(clamp(scene(p), 1e-2, inf)) -- one scene VJP at the surface point;
//   kKind 1, mode 5: the five-tap Laplacian -- the four taps plus the centre.
// Separate instantiations: kinds 1 and 2 add an inlined scene VJP, and kind 3 inside kind 0 cost the training step's
// backward kernel 43 % (116 -> 166 us: registers across the reverse march) -- the per-pixel modes carry none of it.
template <class Cfg, int kKind = 0>
__global__ void __launch_bounds__(256) RM_BWD_OCC k_render_bwd(RenderArgs a) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(a.scene, rm_smem, store, true);
  const int n_acc = Cfg::n_acc(a.scene);
  zero_accumulators<Cfg>(scene, n_acc);
  Tetra T = load_tetra(a.tetra);
  const int W = a.cam.width, H = a.cam.height, rows = a.row_end - a.row_begin;
  const int64_t ntiles = wave_tiles(a);
  Stamps* stamps = nullptr;
#ifdef RM_BWD_STAMPS
  Stamps stamps_;
  stamps_.start();
  stamps = &stamps_;
#endif
  TileCursor tc = first_wave_tile(a, ntiles);
  RM_STAMP(stamps, 7);              // (what lies between the kernel's set-up and its first tile)
  for (; tc.tile < ntiles; ) {
    int cam, row, col;
    bool live = ray_of_lane(a, tc.tile, cam, row, col);
    if (!live) { cam = 0; row = 0; col = 0; }
    int64_t li = ((int64_t)cam * rows + row) * W + col;
    int64_t gi = ((int64_t)cam * H + (row + a.row_begin)) * W + col;
    Pose ps = load_pose(a.orientation, a.translation, cam);
    V3 v = qrot(load3(static_cast<const float*>(a.cam.ray_directions), gi), ps.w, ps.qv);
    V3 p = load3(a.p_final, li);
    V3 gi3 = live ? load3(a.grad_image, li) : mk3(0.0f, 0.0f, 0.0f);
    // the normal for the shader VJP: from the un-normalised normal the recording forward left (same division, same
    // bits), or -- callers without that buffer -- from four tap evaluations
    V3 n, u = mk3(0.0f, 0.0f, 0.0f);
    if (a.normal_u) {
      u = load3(a.normal_u, li);
      const float nu = norm3(u);
      n = mk3(u.x / nu, u.y / nu, u.z / nu);
    } else {
      float lap;
      normals_forward(scene, T, p, 0.0f, n, lap, &u);
    }
    V3 gn = mk3(0.0f, 0.0f, 0.0f);
    V3 gv = mk3(0.0f, 0.0f, 0.0f);
    float gq0 = 0.0f, gq1 = 0.0f, gq2 = 0.0f, gq3 = 0.0f;     // direct dependence of the shader on the pose quaternion
    [[maybe_unused]] V3 gp_direct = mk3(0.0f, 0.0f, 0.0f);   // kind 3: direct dependence of the shader on the origin (- on the surface point)
    if constexpr (kKind == 3) {
      // norm(origin - p).clamp(1e-2, inf).log(): grad / clamped, masked by the clamp, along (origin - p) / norm
      const V3 o = qrot(load3(static_cast<const float*>(a.cam.ray_positions), gi), ps.w, ps.qv) + ps.t;
      const V3 d = o - p;
      const float dn = norm3(d);
      const float gd = (dn >= 1e-2f) ? gi3.x / dn : 0.0f;
      gp_direct = (dn == 0.0f) ? mk3(0.0f, 0.0f, 0.0f) : mk3(gd * (d.x / dn), gd * (d.y / dn), gd * (d.z / dn));
    } else if constexpr (kKind != 0) {
      // nothing here: the upstream of these modes is on the Laplacian / on scene(p), not on the normal (below)
    } else if (a.mode == RM_MODE_LAMBERTIAN) {
      float c = -dot_seq(v, n);
      float g = (c >= 0.0f && c <= 1.0f) ? ((gi3.x + gi3.y) + gi3.z) : 0.0f;   // expand(-1,H,W,3) sums channels
      gn = mk3(-g * v.x, -g * v.y, -g * v.z);
      gv = mk3(-g * n.x, -g * n.y, -g * n.z);                                    // direct dependence of the shader on v
    } else if (a.mode == RM_MODE_NORMAL) {
      gn = mk3((fabsf(n.x) <= 1.0f) ? gi3.x * sgn0(n.x) : 0.0f,
               (fabsf(n.y) <= 1.0f) ? gi3.y * sgn0(n.y) : 0.0f,
               (fabsf(n.z) <= 1.0f) ? gi3.z * sgn0(n.z) : 0.0f);
    } else if (a.mode == RM_MODE_VIGNETTE) {     // (v . col2(q))^3, shader.py:62-66; col2 = third column of QuaternionToSO3
      const float w = ps.w, x = ps.qv.x, y = ps.qv.y, z = ps.qv.z;
      const V3 c2 = mk3(2.0f * (w * y + x * z), 2.0f * (y * z - w * x), ((w * w - x * x) - y * y) + z * z);
      const float d = dot_seq(v, c2);
      const float gd = ((gi3.x + gi3.y) + gi3.z) * (3.0f * (d * d));
      gv = mk3(gd * c2.x, gd * c2.y, gd * c2.z);
      const V3 gc = mk3(gd * v.x, gd * v.y, gd * v.z);
      gq0 = 2.0f * ((gc.x * y - gc.y * x) + gc.z * w);
      gq1 = 2.0f * ((gc.x * z - gc.y * w) - gc.z * x);
      gq2 = 2.0f * ((gc.x * w + gc.y * z) - gc.z * y);
      gq3 = 2.0f * ((gc.x * x + gc.y * y) + gc.z * z);
    } else {   // tangent (6) / spin (7): image = brightness * colormap[index]; the index is piecewise constant
      ShadeIn si;
      si.o = p; si.v = v; si.p = p; si.n = n; si.lap = 0.0f; si.dist = 0.0f; si.qw = ps.w; si.qv = ps.qv; si.col2 = v;
      const Shaded sh = shade_pixel(a.mode, si, a.cmap_size, a.degree);
      float c0, c1, c2;
      if (a.cmap_dtype == RM_DTYPE_F64) {
        const double* c = static_cast<const double*>(a.cmap) + 3 * (int64_t)sh.idx;
        c0 = (float)c[0]; c1 = (float)c[1]; c2 = (float)c[2];
      } else {
        const V3 c = load3_t(a.cmap, sh.idx, a.cmap_dtype);
        c0 = c.x; c1 = c.y; c2 = c.z;
      }
      const float gb = (gi3.x * c0 + gi3.y * c1) + gi3.z * c2;
      // brightness = (re^2 + im^2).pow(1/2): grad * 0.5 / brightness into the sum of squares, 2 x into each term
      const float gs = gb * (0.5f / sh.bright);
      if (a.mode == RM_MODE_TANGENT) {           // shader.py:125-150
        const float c = dot_seq(n, v);
        const V3 tg = mk3((c * v.x) * -1.0f + n.x, (c * v.y) * -1.0f + n.y, (c * v.z) * -1.0f + n.z);
        const V3 u = neg(ps.qv);                 // rotation by conj(q): local = tg + w t + u x t, t = 2 u x tg
        const V3 pr = qrot(tg, ps.w, u);
        const V3 gl = mk3(gs * (2.0f * pr.x), gs * (2.0f * pr.y), 0.0f);
        const V3 t = 2.0f * cross(u, tg);
        const V3 ugl = cross(u, gl);
        const V3 gtg = (gl + 2.0f * cross(u, ugl)) - (2.0f * ps.w) * ugl;
        const V3 gt = ps.w * gl + cross(gl, u);
        const V3 gu = cross(t, gl) + 2.0f * cross(tg, gt);
        gq0 = (gl.x * t.x + gl.y * t.y) + gl.z * t.z;
        gq1 = -gu.x; gq2 = -gu.y; gq3 = -gu.z;
        const float gtv = dot_seq(gtg, v);       // tg = n - (n.v) v
        gn = mk3(gtg.x - gtv * v.x, gtg.y - gtv * v.y, gtg.z - gtv * v.z);
        gv = mk3(-c * gtg.x - gtv * n.x, -c * gtg.y - gtv * n.y, -c * gtg.z - gtv * n.z);
      } else {                                   // shader.py:157-171: r = (0, n) (x) conj(q)
        const float q0 = ps.w, q1 = -ps.qv.x, q2 = -ps.qv.y, q3 = -ps.qv.z;
        const float p1 = n.x, p2 = n.y, p3 = n.z;
        const float r0 = -((p1 * q1 + p2 * q2) + p3 * q3);
        const float r1 = (p1 * q0 + p2 * q3) - p3 * q2;
        const float r2 = (p2 * q0 + p3 * q1) - p1 * q3;
        const float r3 = (p1 * q2 + p3 * q0) - p2 * q1;
        const float re = r0 * r0 - ((r1 * r1 + r2 * r2) + r3 * r3);
        const float nv = norm3(mk3(r1, r2, r3));
        const float im = (nv * r0) * 2.0f;
        const float g_im = gs * (2.0f * im), g_re = gs * (2.0f * re);       // domain_colouring(imag, real): both squared alike
        const float g0 = g_re * (2.0f * r0) + g_im * (2.0f * nv);
        const float sv = (nv == 0.0f) ? 0.0f : (g_im * (2.0f * r0)) / nv;
        const float g1 = g_re * (-2.0f * r1) + sv * r1, g2 = g_re * (-2.0f * r2) + sv * r2, g3 = g_re * (-2.0f * r3) + sv * r3;
        gn = mk3(((-q1 * g0 + q0 * g1) - q3 * g2) + q2 * g3, ((-q2 * g0 + q3 * g1) + q0 * g2) - q1 * g3,
                 ((-q3 * g0 - q2 * g1) + q1 * g2) + q0 * g3);
        const float gc0 = (p1 * g1 + p2 * g2) + p3 * g3;
        const float gc1 = (-p1 * g0 + p3 * g2) - p2 * g3;
        const float gc2 = (-p2 * g0 - p3 * g1) + p1 * g3;
        const float gc3 = (-p3 * g0 + p2 * g1) - p1 * g2;
        gq0 = gc0; gq1 = -gc1; gq2 = -gc2; gq3 = -gc3;
      }
    }
    if (live && a.grad_qdir) {
      float* gq = a.grad_qdir + 4 * li;
      gq[0] = gq0; gq[1] = gq1; gq[2] = gq2; gq[3] = gq3;
    }
    V3 lam = mk3(0.0f, 0.0f, 0.0f);
    int walked = 0;
    bool deferred = false;
    RM_STAMP(stamps, 0);
    if (a.mode != RM_MODE_VIGNETTE) {            // the vignette does not depend on the surface at all
      if constexpr (kKind == 2) {                // proximity: scene(p).clamp(1e-2, inf).log()
        const float dist = scene.eval(p);
        lam = scene.vjp(p, (dist >= 1e-2f) ? gi3.x / dist : 0.0f);
      } else if constexpr (kKind == 3) {
        lam = neg(gp_direct);
      } else {
        lam = normals_backward(scene, T, p, u, gn, kKind == 1 ? gi3.x : 0.0f, kKind == 1);
      }
      RM_STAMP(stamps, 1);
      int ne = a.nexec ? a.nexec[li] : a.steps;
      const int64_t tslot = tc.tile * 64 + (threadIdx.x & 63);
      lam = march_reverse<decltype(scene), DeferToList, true>(scene, lam, v, p, a.traj, a.steps, tslot, ne, a.steps,
                                                              a.grad_dirs != nullptr, gv, a.flags & RM_FLAG_EARLY_OUT, &walked,
                                                              DeferToList{a, li, tslot}, &deferred, stamps);
      RM_STAMP(stamps, 6);
    }
    if (a.tile_cost && (threadIdx.x & 63) == 0) a.tile_cost[tc.tile] = walked;
    if constexpr (kKind == 3) {
      if (live && a.grad_pos) store3(a.grad_pos, li, deferred ? gp_direct : lam + gp_direct);   // (deferred: k_bwd_hard_a adds its part)
    } else {
      if (live && !deferred && a.grad_pos) store3(a.grad_pos, li, lam);
    }
    if (live && !deferred && a.grad_dirs) store3(a.grad_dirs, li, gv);
    RM_STAMP(stamps, 9);            // the tile's stores
    next_wave_tile(a, ntiles, tc);
    RM_STAMP(stamps, 8);            // the next tile (a returning atomic; at the end, the look at all the queues)
  }
#ifdef RM_BWD_STAMPS
  if ((threadIdx.x & 63) == 0 && a.minmax)
    for (int k = 0; k < 10; ++k) atomicAdd(&a.minmax[8 + k], stamps_.acc[k]);
#endif
  flush_accumulators<Cfg>(scene, n_acc, a.partials, rm_smem + ((a.scene.n_params + a.scene.n_derived + 3) & ~3));
}

// ---- deferred rays: every (ray, step) pair in parallel ------------------------------------------------------
// The reverse recursion  lambda_s = lambda_{s+1} + g_s n_s,  g_s = lambda_{s+1}.v,  dL/dtheta += g_s df/dtheta(p_s)
// is serial only in the three numbers lambda; n_s = grad_p f(p_s) and the parameter VJPs do not depend on it.
//   k_bwd_hard_n : n_s (and f(p_s)) for every deferred ray and step        -- parallel over (ray, step)
//   k_bwd_hard_a : the recursion itself, a dozen flops per step             -- parallel over rays
//   k_bwd_hard_b : parameter gradients g_s df/dtheta(p_s)                    -- parallel over (ray, step)
RM_DEV int hard_count(const RenderArgs& a) {
  const uint32_t c = a.minmax[RM_WORK_HARD_COUNT];
  return (int)(c < (uint32_t)a.hard_cap ? c : (uint32_t)a.hard_cap);
}

// wave items: 64 consecutive deferred rays at one step
struct HardItem {
  int s, h;
  bool need;      // this lane has a ray and the step is one of its remaining ones
  int64_t ray;    // band-output index
  int64_t slot;   // trajectory slot
};
RM_DEV HardItem hard_item(const RenderArgs& a, int64_t item, int groups, int H) {
  HardItem it;
  it.s = (int)(item / groups);
  it.h = (int)(item % groups) * 64 + (threadIdx.x & 63);
  const int hc = it.h < H ? it.h : H - 1;
  it.need = it.h < H && it.s <= a.hard_step[hc];
  it.ray = a.hard_ray[hc];
  it.slot = a.hard_slot[hc];
  return it;
}
RM_DEV V3 hard_point(const RenderArgs& a, int s, int64_t ray, int64_t slot) {
  const int ne = a.nexec ? a.nexec[ray] : a.steps;
  return (s < ne) ? traj_load<true>(a.traj, a.steps, slot, s) : load3(a.p_final, ray);
}

template <class Cfg>
__global__ void __launch_bounds__(256) k_bwd_hard_n(RenderArgs a) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(a.scene, rm_smem, store, true);
  const int H = hard_count(a);
  if (H == 0) return;
  const int groups = (H + 63) >> 6;
  const int64_t items = (int64_t)groups * a.steps;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  // Two-deep software pipeline over this wave's items: the list entry (ray, slot, top step) of item + 2 strides and the
  // point of item + 1 stride are in flight while item's VJP runs -- the point's three candidate loads (step count,
  // trajectory, final iterate) are issued together and selected afterwards, so nothing waits on a dependent load
  // (44 % of this kernel's wave-cycles were s_waitcnt with a one-deep prefetch of a three-level chain).
  const int64_t first = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  auto entry = [&](int64_t item) {
    HardItem it{0, 0, false, 0, 0};
    if (item < items) it = hard_item(a, item, groups, H);
    return it;
  };
  auto point = [&](const HardItem& it) {
    const int ne = a.nexec ? a.nexec[it.ray] : a.steps;
    const V3 pt = traj_load<true>(a.traj, a.steps, it.slot, it.s), pf = load3(a.p_final, it.ray);
    return (it.s < ne) ? pt : pf;
  };
  HardItem cur = entry(first), nxt = entry(first + nwaves);
  V3 pc = point(cur);
  for (int64_t item = first; item < items; item += nwaves) {
    const HardItem nn = entry(item + 2 * nwaves);
    const V3 pn = point(nxt);
    if (__any(cur.need)) {
      float f;
      const V3 n = scene.vjp_point(pc, 1.0f, &f);
      if (cur.need) {
        const int64_t at = 4 * ((int64_t)cur.s * a.hard_cap + cur.h);
        *reinterpret_cast<float4*>(a.hard_n + at) = make_float4(n.x, n.y, n.z, f);
        // the point itself, next to its gradient: k_bwd_hard_b addresses both by the pair code alone
        *reinterpret_cast<float4*>(a.hard_p + at) = make_float4(pc.x, pc.y, pc.z, 0.0f);
      }
    }
    cur = nxt; pc = pn; nxt = nn;
  }
}

__global__ void k_bwd_hard_a(RenderArgs a) {
  const int H = hard_count(a);
  const int h0 = blockIdx.x * blockDim.x + threadIdx.x;
  if ((h0 & ~63) >= H) return;                    // whole waves leave; the last wave keeps its idle lanes (ballots below)
  const bool valid = h0 < H;
  const int h = valid ? h0 : H - 1;
  const int W = a.cam.width, Hh = a.cam.height, rows = a.row_end - a.row_begin;
  const int64_t li = a.hard_ray[h], tslot = a.hard_slot[h];
  const int cam = (int)(li / ((int64_t)rows * W));
  const int64_t rem = li - (int64_t)cam * rows * W;
  const int row = (int)(rem / W), col = (int)(rem - (int64_t)row * W);
  const int64_t gi = ((int64_t)cam * Hh + (row + a.row_begin)) * W + col;
  const Pose ps = load_pose(a.orientation, a.translation, cam);
  const V3 v = qrot(load3(static_cast<const float*>(a.cam.ray_directions), gi), ps.w, ps.qv);
  const float* st = a.hard_state + 8 * (int64_t)h;
  V3 lam = mk3(st[0], st[1], st[2]), gv = mk3(st[3], st[4], st[5]);
  const bool want_gv = a.grad_dirs != nullptr;
  const int top = valid ? a.hard_step[h] : -1;
  int s = top;
  bool frozen = false;
  while (s >= 0 && !frozen) {
    // the recursion is a chain of dependent flops, the n_s it consumes are not: fetch eight steps' worth at once
    float4 nf[8];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      nf[k] = (s - k >= 0) ? *reinterpret_cast<const float4*>(a.hard_n + 4 * ((int64_t)(s - k) * a.hard_cap + h))
                           : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (s < 0 || frozen) break;
      const float gf = (lam.x * v.x + lam.y * v.y) + lam.z * v.z;
      if (fabsf(gf) <= 2.4e-7f * ((fabsf(lam.x * v.x) + fabsf(lam.y * v.y)) + fabsf(lam.z * v.z))) { frozen = true; break; }   // noise
      a.hard_g[(int64_t)s * a.hard_cap + h] = gf;
      if (want_gv) gv = gv + nf[k].w * lam;
      lam = lam + mk3(gf * nf[k].x, gf * nf[k].y, gf * nf[k].z);
      --s;
    }
  }
  // The (ray, step) pairs k_bwd_hard_b has to visit are this ray's steps top .. s+1: listed densely, the space for
  // a whole wave reserved with ONE returning atomic (a same-address atomic per step cost 12-16 ns each,
  // chip-wide: 220 us for this kernel when it was tried).
  {
    const int lane = threadIdx.x & 63;
    const uint32_t cnt = valid ? (uint32_t)(top - s) : 0u;
    uint32_t incl = cnt;
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o, 64);
      if (lane >= o) incl += up;
    }
    uint32_t base = 0;
    if (lane == 63) base = atomicAdd(&a.minmax[RM_WORK_HARD_PAIRS], incl);
    base = (uint32_t)__shfl((int)base, 63, 64) + (incl - cnt);
    for (uint32_t k = 0; k < cnt; ++k)
      a.hard_pairs[base + k] = ((uint32_t)h << RM_HARD_STEP_BITS) | (uint32_t)(top - (int)k);
  }
  if (want_gv && s >= 0) {     // steps 0..s with lambda frozen: sum_i f(p_i) = (p_{s+1} - p_0).v / |v|^2
    const V3 dp = hard_point(a, s + 1, li, tslot) - traj_load<true>(a.traj, a.steps, tslot, 0);
    const float sumf = ((dp.x * v.x + dp.y * v.y) + dp.z * v.z) / ((v.x * v.x + v.y * v.y) + v.z * v.z);
    gv = gv + sumf * lam;
  }
  // (distance shader: k_render_bwd left the shader's direct dependence on the origin there for the deferred rays)
  if (valid && a.grad_pos) store3(a.grad_pos, li, a.mode == RM_MODE_DISTANCE ? lam + load3(a.grad_pos, li) : lam);
  if (valid && a.grad_dirs) store3(a.grad_dirs, li, gv);
}

template <class Cfg>
__global__ void __launch_bounds__(256) RM_HARDB_OCC k_bwd_hard_b(RenderArgs a) {
  typename Cfg::Store store;
  auto scene = Cfg::setup(a.scene, rm_smem, store, true);
  const int n_acc = Cfg::n_acc(a.scene);
  zero_accumulators<Cfg>(scene, n_acc);
  // the (ray, step) pairs with a non-zero upstream, listed densely by k_bwd_hard_a: 64 of them per wave item,
  // whatever ray and step they belong to (config 4: 10.8 k items instead of 17.6 k for ray-aligned items).  A pair's
  // upstream and point are both addressed by its code alone (k_bwd_hard_n left the point next to its gradient):
  // one level of indirection instead of code -> ray -> nexec -> trajectory
  const int64_t npairs = a.minmax[RM_WORK_HARD_PAIRS];
  const int64_t items = (npairs + 63) >> 6;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t first = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  // two-deep pipeline: the pair code of item + 2 strides and the (upstream, point) of item + 1 stride are in flight
  // while item's VJP runs
  auto code_of = [&](int64_t item) -> uint32_t {
    const int64_t k = item * 64 + (threadIdx.x & 63);
    return (item < items) ? a.hard_pairs[k < npairs ? k : npairs - 1] : 0u;
  };
  auto data_of = [&](int64_t item, uint32_t code, float& g, V3& p) {
    const int64_t k = item * 64 + (threadIdx.x & 63);
    const int h = (int)(code >> RM_HARD_STEP_BITS), st = (int)(code & ((1u << RM_HARD_STEP_BITS) - 1u));
    const int64_t at = (int64_t)st * a.hard_cap + h;
    g = 0.0f; p = mk3(0.0f, 0.0f, 0.0f);
    if (item >= items) return;
    const float4 pt = *reinterpret_cast<const float4*>(a.hard_p + 4 * at);
    g = (k < npairs) ? a.hard_g[at] : 0.0f;
    p = mk3(pt.x, pt.y, pt.z);
  };
  uint32_t code1 = code_of(first + nwaves);
  float gc; V3 pc;
  data_of(first, code_of(first), gc, pc);
  for (int64_t item = first; item < items; item += nwaves) {
    const uint32_t code2 = code_of(item + 2 * nwaves);
    float gn; V3 pn;
    data_of(item + nwaves, code1, gn, pn);
    scene.vjp(pc, gc);
    gc = gn; pc = pn; code1 = code2;
  }
  flush_accumulators<Cfg>(scene, n_acc, a.partials, rm_smem + ((a.scene.n_params + a.scene.n_derived + 3) & ~3));
}

// Gradient reduction, deterministic (fixed summation tree, no float atomics).
// Stage 1: one block per accumulator; thread t sums rows t, t+256, ... then a fixed LDS tree.
__global__ void k_reduce_partials(const float* __restrict__ partials, int nblocks, int n_acc,
                                  float* __restrict__ sums) {
  __shared__ float red[256];
  const int i = blockIdx.x;
  float acc = 0.0f;
  for (int b = threadIdx.x; b < nblocks; b += 256) acc += partials[(int64_t)b * n_acc + i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[i] = red[0];
}

// Stage 2: push the gradients of the derived capsule constants back onto start/end
// (AB = end - start, ABs = AB / |AB|^2; primitives.py:52-54) and emit the raw-parameter vector.
__global__ void k_finish_grads(RmScene sc, const float* __restrict__ sums, float* __restrict__ grad_params) {
  extern __shared__ float s_acc[];
  const int n_acc = sc.n_params + sc.n_grad_derived;
  for (int i = threadIdx.x; i < n_acc; i += blockDim.x) s_acc[i] = sums[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    const int4* prog = reinterpret_cast<const int4*>(sc.program);
    for (int pc = 0; pc < sc.n_instr; ++pc) {
      int4 w = prog[pc];
      if (w.x != RM_OP_LINE) continue;
      float a[6];
      for (int k = 0; k < 6; ++k) {
        if (sc.block) a[k] = sc.block[w.y + k];
        else if (sc.param_refs) { const RmParamRef r = sc.param_refs[w.y + k]; a[k] = ld_t(r.base, r.elem, r.dtype); }
        else a[k] = sc.params[w.y + k];
      }
      float ab[3] = {a[3] - a[0], a[4] - a[1], a[5] - a[2]};
      float len2 = (ab[0] * ab[0] + ab[1] * ab[1]) + ab[2] * ab[2];
      const float* gab = s_acc + w.z;
      const float* gabs = s_acc + w.z + 3;
      // ABs = AB / len2 : g_AB += gABs/len2 ; g_len2 = -sum(gABs * AB)/len2^2 ; len2 = sum AB^2
      float glen2 = -((gabs[0] * ab[0] + gabs[1] * ab[1]) + gabs[2] * ab[2]) / (len2 * len2);
      for (int k = 0; k < 3; ++k) {
        float g = gab[k] + gabs[k] / len2 + 2.0f * ab[k] * glen2;
        s_acc[w.y + 3 + k] += g;   // end
        s_acc[w.y + k] -= g;       // start
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < sc.n_params; i += blockDim.x) grad_params[i] = s_acc[i];
}

}  // namespace rm

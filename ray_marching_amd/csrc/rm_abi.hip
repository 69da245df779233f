// rm_abi.hip -- C ABI (include/rm_abi.h) over the generic interpreter kernels.
// Built for gfx950 only:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC rm_abi.hip -o librm_hip.so
//
// Second build mode: -DRM_STATIC_CODE='"code_<hash>.h"' bakes one scene program in at compile
// time (rm::StaticCfg): same entry points, same handlers, but the interpreter loop, the LDS
// stack/tape and the scalar branches are gone (ray_marching_amd/specialize.py drives this).
#include "rm_kernels.h"

#ifdef RM_STATIC_CODE
#include RM_STATIC_CODE
#endif

#include <atomic>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

constexpr int kMaxBlocks = 2048;        // persistent grid cap: 256 CUs x 8 blocks
constexpr int kMaxBlocksBwd = 2048;     // rows of per-block partial sums a backward kernel may write
constexpr int kHardBlocksB = 2048;      // + rows of k_bwd_hard_b (deferred rays)
constexpr int kPartialRows = kMaxBlocksBwd + kHardBlocksB;
constexpr size_t kLdsDefault = 64 * 1024;
constexpr size_t kLdsMax = 160 * 1024;  // gfx950: 160 KiB per CU

int check_static(const RmScene* sc);

int check_scene(const RmScene* sc) {
  if (!sc || !sc->program || (!sc->params && !sc->param_refs && !sc->block && sc->n_params > 0))
    return fail(RM_E_BADARG, "scene: null program/params");
  if (sc->n_instr <= 0 || sc->n_instr > 4096 || sc->n_params < 0 || sc->n_derived < 0 ||
      sc->stack_floats < 0 || sc->n_slots < 0 || sc->n_grad_derived < 0 || sc->n_grad_derived > sc->n_derived)
    return fail(RM_E_BADARG, "scene: bad sizes (n_instr=%d n_params=%d)", sc->n_instr, sc->n_params);
  return check_static(sc);
}

bool io_dtype_ok(int dt) { return dt == RM_DTYPE_F32 || dt == RM_DTYPE_F16; }

struct Launch {
  int block;
  size_t lds;
};

#ifdef RM_STATIC_CODE
#ifndef RM_FWD_REG_PARAMS
#define RM_FWD_REG_PARAMS 32      // forward kernels: parameter floats hoisted into SGPRs up to this many, LDS reads beyond (closed scene 1, 51 floats:
                                  // 225 SGPR spills as registers; training step 0.401 -> 0.394 ms at 512^2, 0.878 -> 0.838 at 1024^2, profiles/r03_train_ab.txt)
#endif
using G = rm::StaticCfg<RmStaticCode, RM_FWD_REG_PARAMS>;
#ifndef RM_BWD_REG_PARAMS
#define RM_BWD_REG_PARAMS 64
#endif
using GB = rm::StaticCfg<RmStaticCode, RM_BWD_REG_PARAMS>;   // backward kernels: parameter floats kept in SGPRs up to this many
// the tile frame kernel of scenes with few parameters keeps them in VGPRs (rm_device.h: RegParams)
#ifndef RM_FWD_VGPR_PARAM_LIMIT
#define RM_FWD_VGPR_PARAM_LIMIT 0      // off: see RegParams (rm_device.h) -- it won while one kernel instantiation also carried the
#endif                                 // recording code (997 v_readlane); with the inference instantiation split off the SGPR form is ahead
using GF = rm::StaticCfg<RmStaticCode, RM_FWD_REG_PARAMS, (RmStaticCode::n_params + RmStaticCode::n_derived <= RM_FWD_VGPR_PARAM_LIMIT)>;
// static path: parameter block + (backward) one accumulator row per wave for the block reduction
size_t lds_bytes(const RmScene& sc, int block, bool backward) {
  size_t pb = (size_t)((sc.n_params + sc.n_derived + 3) & ~3);
  return 4 * (pb + (backward ? (size_t)(block >> 6) * (sc.n_params + sc.n_grad_derived) : 0) + 4 + (size_t)G::kTapeFloats);
}
int check_static(const RmScene* sc) {
  if (sc->n_instr != RmStaticCode::n || sc->n_params != RmStaticCode::n_params ||
      sc->n_derived != RmStaticCode::n_derived || sc->n_slots != RmStaticCode::n_slots ||
      sc->n_grad_derived != RmStaticCode::n_grad_derived)
    return fail(RM_E_PROGRAM, "scene does not match the program this library was specialised for");
  return RM_OK;
}
#else
using G = rm::GenericCfg;
using GB = rm::GenericCfg;
using GF = rm::GenericCfg;
int check_static(const RmScene*) { return RM_OK; }
// LDS bytes of the generic path for a block of `block` threads.
size_t lds_bytes(const RmScene& sc, int block, bool backward) {
  return 4 * rm::GenericCfg::lds_floats(sc, block, backward);
}
#endif

template <class K>
int pick_launch(K kernel, const RmScene& sc, bool backward, int max_block, Launch* out) {
  for (int block = max_block; block >= 64; block >>= 1) {
    size_t b = lds_bytes(sc, block, backward);
    if (b <= kLdsDefault) { *out = {block, b}; return RM_OK; }
  }
  size_t b = lds_bytes(sc, 64, backward);
  if (b > kLdsMax) return fail(RM_E_TOO_LARGE, "scene needs %zu B of LDS per 64-ray block (max %zu)", b, kLdsMax);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)b);
  if (e != hipSuccess) return fail(RM_E_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  *out = {64, b};
  return RM_OK;
}

int grid_for(int64_t tiles, int cap) {
  if (tiles < 1) tiles = 1;
  return (int)(tiles < cap ? tiles : cap);
}

// tuning knobs (environment, read once): RM_BLOCK = threads per block of the frame kernels,
// RM_MAX_BLOCKS = persistent-grid cap (0 = one block per 'blockDim/64' wave tiles, i.e. no
// persistent loop: the hardware workgroup dispatcher hands out the tiles).
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
bool env_set(const char* name) { const char* v = getenv(name); return v && *v; }
// per-device caches (a process may render on several devices; HIP function attributes and the CU count are per
// device): slot = the current device's ordinal, atomics because two host threads may race on the first use
constexpr int RM_MAX_DEVICES = 64;
int current_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= RM_MAX_DEVICES) return -1;
  return dev;
}
int cu_count() {
  static std::atomic<int> cache[RM_MAX_DEVICES];
  const int slot = current_device_slot();
  int n = slot >= 0 ? cache[slot].load(std::memory_order_relaxed) : 0;
  if (n > 0) return n;
  int cus = 256;
  if (slot >= 0) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, slot);
  if (cus <= 0) cus = 256;
  if (slot >= 0) cache[slot].store(cus, std::memory_order_relaxed);
  return cus;
}
int tune_block() { static int v = env_int("RM_BLOCK", 256); return v; }
// frame kernel: 1280 blocks x 4 waves = 5 waves/SIMD measured best with the atomic tile queues
// (profiles/grid_sweep.py: 512 -> 463 us, 1024 -> 376, 1280 -> 371, 2048 -> 390)
int tune_max_blocks() { static int v = env_int("RM_MAX_BLOCKS", 1280); return v; }
// backward kernels: blocks of 128 threads; 1024 = 2 waves per SIMD (what 250-VGPR kernels can hold)
int tune_bwd_blocks() { static int v = env_int("RM_BWD_BLOCKS", 1024); return v < 1 ? 1 : (v > kMaxBlocksBwd ? kMaxBlocksBwd : v); }
// k_bwd_hard_b holds 3 waves per SIMD (155 VGPRs): 1536 blocks of 2 waves fill them (1024 left a slot idle: 0.767 -> 0.755 ms per
// config-4 step at 1024^2, 0.328 -> 0.326 at 512^2, once the scene block made blocks cheap; profiles/r03_train_ab.txt)
int tune_hardb_blocks() { static int v = env_int("RM_HARDB_BLOCKS", 1536); return v < 1 ? 1 : (v > kHardBlocksB ? kHardBlocksB : v); }
int tune_hardn_blocks() { static int v = env_int("RM_HARDN_BLOCKS", 2048); return v < 1 ? 1 : v; }

int launched(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RM_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return RM_OK;
}

// partials holds kPartialRows rows of per-block sums followed by one row of totals
int reduce_partials(const RmScene& sc, float* partials, int nblocks, float* grad_params, hipStream_t s) {
  if (!grad_params) return RM_OK;
  const int n_acc = sc.n_params + sc.n_grad_derived;
  if (n_acc == 0) return RM_OK;
  float* sums = partials + (size_t)kPartialRows * n_acc;
  rm::k_reduce_partials<<<n_acc, 256, 0, s>>>(partials, nblocks, n_acc, sums);
  if (int e = launched("k_reduce_partials")) return e;
  size_t lds = 4 * (size_t)(n_acc + 1);
  rm::k_finish_grads<<<1, 256, lds, s>>>(sc, sums, grad_params);
  return launched("k_finish_grads");
}

}  // namespace

extern "C" {

int rm_abi_version(void) { return RM_ABI_VERSION; }

const char* rm_last_error(void) { return g_err; }

int64_t rm_grad_partials_floats(const RmScene* scene, int64_t n) {
  if (!scene) return 0;
  (void)n;
  return (int64_t)(kPartialRows + 1) * (scene->n_params + scene->n_grad_derived);
}

int64_t rm_bwd_hard_floats(int64_t capacity, int32_t steps) {
  if (capacity <= 0 || steps < 0) return 0;
  // ray, trajectory slot, step, state[8]; n[steps][cap][4], p[steps][cap][4], g[steps][cap], pairs[steps*cap]
  return capacity * (3 + 8) + (int64_t)steps * capacity * 10;
}

int rm_sdf_forward(const RmScene* scene, const void* points, void* dist, int64_t n, int32_t dtype, void* stream) {
  if (int e = check_scene(scene)) return e;
  if (n < 0 || (n > 0 && (!points || !dist))) return fail(RM_E_BADARG, "rm_sdf_forward: null buffer");
  if (!io_dtype_ok(dtype)) return fail(RM_E_BADARG, "rm_sdf_forward: dtype %d is neither F32 nor F16", dtype);
  if (n == 0) return RM_OK;
  Launch L;
  if (int e = pick_launch(rm::k_sdf_fwd<G>, *scene, false, 256, &L)) return e;
  int grid = grid_for((n + L.block - 1) / L.block, kMaxBlocks);
  rm::k_sdf_fwd<G><<<grid, L.block, L.lds, (hipStream_t)stream>>>(*scene, points, dist, n, dtype);
  return launched("k_sdf_fwd");
}

int rm_sdf_backward(const RmScene* scene, const float* points, const float* grad_dist, float* grad_points,
                    float* grad_params, float* partials, int64_t n, void* stream) {
#ifdef RM_NO_BACKWARD
  return fail(RM_E_BADARG, "rm_sdf_backward: this specialised library was built forward-only");
#else
  if (int e = check_scene(scene)) return e;
  if (n <= 0 || !points || !grad_dist || !partials) return fail(RM_E_BADARG, "rm_sdf_backward: null buffer / n<=0");
  Launch L;
  if (int e = pick_launch(rm::k_sdf_bwd<GB>, *scene, true, 128, &L)) return e;
  int grid = grid_for((n + L.block - 1) / L.block, 1024);
  rm::k_sdf_bwd<GB><<<grid, L.block, L.lds, (hipStream_t)stream>>>(*scene, points, grad_dist, grad_points, partials, n);
  if (int e = launched("k_sdf_bwd")) return e;
  return reduce_partials(*scene, partials, grid, grad_params, (hipStream_t)stream);
#endif
}

int rm_march_forward(const RmScene* scene, const void* pos, const void* dirs, void* out_pos, float* traj,
                     int32_t* nexec, int64_t n, int32_t steps, int32_t flags, int32_t dtype, void* stream) {
  if (int e = check_scene(scene)) return e;
  if (n < 0 || steps < 0 || (n > 0 && (!pos || !dirs || !out_pos))) return fail(RM_E_BADARG, "rm_march_forward: bad args");
  if (!io_dtype_ok(dtype)) return fail(RM_E_BADARG, "rm_march_forward: dtype %d is neither F32 nor F16", dtype);
  if (n == 0) return RM_OK;
  Launch L;
  if (int e = pick_launch(rm::k_march_fwd<G>, *scene, false, 256, &L)) return e;
  int grid = grid_for((n + L.block - 1) / L.block, kMaxBlocks);
  rm::k_march_fwd<G><<<grid, L.block, L.lds, (hipStream_t)stream>>>(*scene, pos, dirs, out_pos, traj, nexec, n, steps, flags, dtype);
  return launched("k_march_fwd");
}

int rm_march_backward(const RmScene* scene, const float* dirs, const float* traj, const int32_t* nexec,
                      const float* grad_out, float* grad_pos, float* grad_dirs, float* grad_params,
                      float* partials, int64_t n, int32_t steps, void* stream) {
#ifdef RM_NO_BACKWARD
  return fail(RM_E_BADARG, "rm_march_backward: this specialised library was built forward-only");
#else
  if (int e = check_scene(scene)) return e;
  if (n <= 0 || steps < 0 || !dirs || !grad_out || !partials || (steps > 0 && !traj))
    return fail(RM_E_BADARG, "rm_march_backward: bad args");
  Launch L;
  if (int e = pick_launch(rm::k_march_bwd<GB>, *scene, true, 128, &L)) return e;
  int grid = grid_for((n + L.block - 1) / L.block, 1024);
  rm::k_march_bwd<GB><<<grid, L.block, L.lds, (hipStream_t)stream>>>(*scene, dirs, traj, nexec, grad_out, grad_pos, grad_dirs,
                                                                   partials, n, steps);
  if (int e = launched("k_march_bwd")) return e;
  return reduce_partials(*scene, partials, grid, grad_params, (hipStream_t)stream);
#endif
}

int rm_normals_forward(const RmScene* scene, const RmTetra* tetra, const void* coords, void* normals,
                       void* laplacian, int64_t n, int32_t dtype, void* stream) {
  if (int e = check_scene(scene)) return e;
  if (!tetra || n < 0 || (n > 0 && (!coords || !normals || !laplacian))) return fail(RM_E_BADARG, "rm_normals_forward: bad args");
  if (!io_dtype_ok(dtype)) return fail(RM_E_BADARG, "rm_normals_forward: dtype %d is neither F32 nor F16", dtype);
  if (n == 0) return RM_OK;
  Launch L;
  if (int e = pick_launch(rm::k_normals_fwd<G>, *scene, false, 256, &L)) return e;
  int grid = grid_for((n + L.block - 1) / L.block, kMaxBlocks);
  rm::k_normals_fwd<G><<<grid, L.block, L.lds, (hipStream_t)stream>>>(*scene, *tetra, coords, normals, laplacian, n, dtype);
  return launched("k_normals_fwd");
}

int rm_normals_backward(const RmScene* scene, const RmTetra* tetra, const float* coords, const float* grad_normals,
                        const float* grad_lap, float* grad_coords, float* grad_params, float* partials, int64_t n,
                        void* stream) {
#ifdef RM_NO_BACKWARD
  return fail(RM_E_BADARG, "rm_normals_backward: this specialised library was built forward-only");
#else
  if (int e = check_scene(scene)) return e;
  if (!tetra || n <= 0 || !coords || !partials) return fail(RM_E_BADARG, "rm_normals_backward: bad args");
  Launch L;
  if (int e = pick_launch(rm::k_normals_bwd<GB>, *scene, true, 128, &L)) return e;
  int grid = grid_for((n + L.block - 1) / L.block, 1024);
  rm::k_normals_bwd<GB><<<grid, L.block, L.lds, (hipStream_t)stream>>>(*scene, *tetra, coords, grad_normals, grad_lap,
                                                                     grad_coords, partials, n);
  if (int e = launched("k_normals_bwd")) return e;
  return reduce_partials(*scene, partials, grid, grad_params, (hipStream_t)stream);
#endif
}

int rm_camera_forward(const RmCamera* cam, const void* orientation, const void* translation, void* out_pos,
                      void* out_dirs, void* out_frames, void* stream) {
  if (!cam || !cam->ray_positions || !cam->ray_directions || !orientation || !translation || !out_pos || !out_dirs)
    return fail(RM_E_BADARG, "rm_camera_forward: null buffer");
  if (!io_dtype_ok(cam->dtype)) return fail(RM_E_BADARG, "rm_camera_forward: camera dtype %d is neither F32 nor F16", cam->dtype);
  if (cam->num_cameras <= 0 || cam->num_cameras > 256 || cam->height <= 0 || cam->width <= 0)
    return fail(RM_E_BADARG, "rm_camera_forward: bad camera shape");
  int64_t n = (int64_t)cam->num_cameras * cam->height * cam->width;
  int grid = grid_for((n + 255) / 256, kMaxBlocks);
  rm::k_camera_fwd<<<grid, 256, 0, (hipStream_t)stream>>>(*cam, orientation, translation, out_pos, out_dirs, out_frames);
  return launched("k_camera_fwd");
}

static int check_render(const RmScene* scene, const RmCamera* cam, const RmTetra* tetra, const void* orientation,
                        const void* translation, int32_t steps, int32_t row_begin, int32_t row_end) {
  if (int e = check_scene(scene)) return e;
  if (!cam || !tetra || !cam->ray_positions || !cam->ray_directions || !orientation || !translation)
    return fail(RM_E_BADARG, "render: null camera / pose");
  if (!io_dtype_ok(cam->dtype)) return fail(RM_E_BADARG, "render: camera dtype %d is neither F32 nor F16", cam->dtype);
  if (cam->num_cameras <= 0 || cam->height <= 0 || cam->width <= 0) return fail(RM_E_BADARG, "render: bad camera shape");
  if (steps < 0 || row_begin < 0 || row_end > cam->height || row_begin >= row_end)
    return fail(RM_E_BADARG, "render: bad steps/rows (%d, [%d,%d) of %d)", steps, row_begin, row_end, cam->height);
  return RM_OK;
}

static int64_t wave_tile_count(int32_t num_cameras, int32_t rows, int32_t width, int32_t flags) {
  return (flags & RM_FLAG_TILE8X8) ? (int64_t)num_cameras * ((width + 7) >> 3) * ((rows + 7) >> 3)
                                   : ((int64_t)num_cameras * rows * width + 63) / 64;
}

int64_t rm_wave_tiles(int32_t num_cameras, int32_t rows, int32_t width, int32_t flags) {
  if (num_cameras <= 0 || rows <= 0 || width <= 0) return 0;
  return wave_tile_count(num_cameras, rows, width, flags);
}

int rm_render_forward(const RmScene* scene, const RmCamera* cam, const RmTetra* tetra, const void* orientation,
                      const void* translation, void* image, int32_t image_dtype, float* first_pass, float* p_final,
                      float* traj, int32_t* nexec, float* normal_u, uint32_t* minmax, const void* cmap, int32_t cmap_size,
                      int32_t cmap_dtype, int32_t mode, int32_t degree, int32_t steps, int32_t row_begin,
                      int32_t row_end, int32_t flags, const int32_t* tile_order, int32_t* tile_cost,
                      float* park_ws, int64_t park_capacity, void* stream) {
  if (int e = check_render(scene, cam, tetra, orientation, translation, steps, row_begin, row_end)) return e;
  if (!image) return fail(RM_E_BADARG, "rm_render_forward: null image");
  if (mode < 0 || mode > 7) return fail(RM_E_BADARG, "rm_render_forward: mode %d not in 0..7", mode);
  const bool global = (mode == RM_MODE_DISTANCE || mode == RM_MODE_PROXIMITY || mode == RM_MODE_LAPLACIAN);
  const bool mapped = (mode == RM_MODE_TANGENT || mode == RM_MODE_SPIN);
  if (global && (!minmax || !first_pass))
    return fail(RM_E_BADARG, "rm_render_forward: mode %d needs a minmax workspace and a first_pass buffer", mode);
  if (mapped && (!cmap || cmap_size <= 0 || cmap_dtype < RM_DTYPE_F32 || cmap_dtype > RM_DTYPE_F64))
    return fail(RM_E_BADARG, "rm_render_forward: mode %d needs a colormap (F32, F16 or F64)", mode);
  if (!(io_dtype_ok(image_dtype) || (image_dtype == RM_DTYPE_F64 && mapped) || (image_dtype == RM_DTYPE_RGBA_F32 && cam->num_cameras == 1)))
    return fail(RM_E_BADARG, "rm_render_forward: image dtype %d (F64 only for modes 6, 7; RGBA_F32 only for one camera)", image_dtype);
  if (image_dtype == RM_DTYPE_RGBA_F32 && (reinterpret_cast<uintptr_t>(image) & 15))
    return fail(RM_E_BADARG, "rm_render_forward: an RGBA_F32 image must be 16-byte aligned");
  rm::RenderArgs a;
  memset(&a, 0, sizeof(a));
  a.scene = *scene; a.cam = *cam; a.tetra = *tetra;
  a.orientation = orientation; a.translation = translation;
  a.image = image; a.image_dtype = image_dtype; a.first_pass = first_pass;
  a.p_final = p_final; a.traj = traj; a.nexec = nexec; a.normal_u = normal_u; a.minmax = minmax;
  a.cmap = cmap; a.cmap_size = cmap_size; a.cmap_dtype = cmap_dtype;
  a.mode = mode; a.degree = degree; a.steps = steps; a.row_begin = row_begin; a.row_end = row_end; a.flags = flags;
  a.tile_order = tile_order; a.tile_cost = tile_cost;
  // ray parking (include/rm_abi.h): dense second kernel for the minority of rays that never settle
  const int64_t park_seg = park_capacity / (RM_PARK_LISTS * RM_PARK_SHARDS);
#ifdef RM_PARKING
  const bool park = park_ws && park_seg >= 64 && park_seg < ((int64_t)1 << 30) && minmax && !traj &&
                    (flags & RM_FLAG_EARLY_OUT) && steps >= 48;
#else
  const bool park = false;      // this library was built without -DRM_PARKING: the workspace is ignored
#endif
  if (park) {
    a.park_seg = (int32_t)park_seg;
    a.park_ray = reinterpret_cast<int32_t*>(park_ws);
    a.park_p = park_ws + park_seg * RM_PARK_LISTS * RM_PARK_SHARDS;
  }
  const int64_t wave_tiles = wave_tile_count(cam->num_cameras, row_end - row_begin, cam->width, flags);
  if (flags & RM_FLAG_REGEN) {
    if (!(flags & RM_FLAG_EARLY_OUT) || !(flags & RM_FLAG_TILE8X8) || !minmax || !p_final || (steps & 3) || traj || nexec || normal_u || park)
      return fail(RM_E_BADARG, "rm_render_forward: RM_FLAG_REGEN needs EARLY_OUT, TILE8X8, minmax, p_final, steps %% 4 == 0 "
                               "and no traj / nexec / parking");
    if (wave_tiles * 64 >= ((int64_t)1 << 31)) return fail(RM_E_BADARG, "rm_render_forward: RM_FLAG_REGEN: too many rays");
    if (tile_cost && hipMemsetAsync(tile_cost, 0, sizeof(int32_t) * wave_tiles * 64, (hipStream_t)stream) != hipSuccess)
      return fail(RM_E_LAUNCH, "rm_render_forward: clearing tile_cost failed");
    Launch LM, LF;
    if (int e = pick_launch(rm::k_march_regen<G>, *scene, false, tune_block(), &LM)) return e;
    if (int e = pick_launch(rm::k_render_finish<G>, *scene, false, tune_block(), &LF)) return e;
    // persistent pools: as many blocks as the chip holds at once, never more than there are tiles to start with
    int per_cu = 0;
    int gm = 4 * cu_count();
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rm::k_march_regen<G>, LM.block, LM.lds) == hipSuccess && per_cu > 0)
      gm = per_cu * cu_count();
    // (measured at 1080p, profiles/regen_probe.py: 7 blocks per CU 350 us, 5 -> 319, 4 -> 321, 3 -> 338: more pools in
    // flight means more lanes left idle once the queues are dry)
    if (gm > 5 * cu_count()) gm = 5 * cu_count();
    if (env_set("RM_MAX_BLOCKS") && tune_max_blocks() > 0) gm = tune_max_blocks();
    const int64_t tile_blocks = (wave_tiles + (LM.block >> 6) - 1) / (LM.block >> 6);
    if (gm > tile_blocks) gm = (int)(tile_blocks < 1 ? 1 : tile_blocks);
    rm::k_march_regen<G><<<gm, LM.block, LM.lds, (hipStream_t)stream>>>(a);
    if (int e = launched("k_march_regen")) return e;
    int gf = grid_for((wave_tiles + (LF.block >> 6) - 1) / (LF.block >> 6), tune_max_blocks() > 0 ? tune_max_blocks() : kMaxBlocks);
    // the second kernel takes the finished scene block the first one left (RmScene::block_out), when the caller gave room for it
    rm::RenderArgs af = a;
    if (a.scene.block_out && !a.scene.block) { af.scene.block = a.scene.block_out; af.scene.block_out = nullptr; }
    rm::k_render_finish<G><<<gf, LF.block, LF.lds, (hipStream_t)stream>>>(af);
    return launched("k_render_finish");
  }
  // two instantiations: the plain inference frame, and the one that also records (trajectory / p_final / nexec / normal_u)
  const bool rec = traj || p_final || nexec || normal_u;
  Launch L;
  if (int e = rec ? pick_launch(rm::k_render_fwd<GF, true>, *scene, false, tune_block(), &L)
                  : pick_launch(rm::k_render_fwd<GF, false>, *scene, false, tune_block(), &L)) return e;
  int64_t tiles = (wave_tiles + (L.block >> 6) - 1) / (L.block >> 6);
  int grid = tune_max_blocks() > 0 ? grid_for(tiles, tune_max_blocks()) : (int)tiles;
  if (tune_max_blocks() > 0 && (flags & RM_FLAG_DYNAMIC_TILES) && minmax && !env_set("RM_MAX_BLOCKS")) {
    // Persistent grid, two more limits (measured, profiles/ab_probe.py): (1) never more blocks than the chip
    // holds at once -- a block that starts late pays the scene staging for an empty queue (the 32-primitive
    // kernel is register-limited to fewer than 5 waves per SIMD: 1280 -> 512 blocks, 14.2 -> 13.05 ms per 8K
    // band); (2) at least ~2 tiles per wave, or the dynamic queues have nothing to balance with (512^2 frame,
    // closed scene 1: 1280 -> 512 blocks, 228 -> 177 us).
    int per_cu = 0;
    const hipError_t oe = rec ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rm::k_render_fwd<GF, true>, L.block, L.lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rm::k_render_fwd<GF, false>, L.block, L.lds);
    if (oe == hipSuccess && per_cu > 0) {
      const int resident = per_cu * cu_count();
      if (grid > resident) grid = resident;
    }
    const int64_t two_per_wave = (wave_tiles / 2 + (L.block >> 6) - 1) / (L.block >> 6);
    if (grid > two_per_wave) grid = (int)(two_per_wave < 1 ? 1 : two_per_wave);
  }
  if (rec) rm::k_render_fwd<GF, true><<<grid, L.block, L.lds, (hipStream_t)stream>>>(a);
  else rm::k_render_fwd<GF, false><<<grid, L.block, L.lds, (hipStream_t)stream>>>(a);
  if (int e = launched("k_render_fwd")) return e;
  if (park) {
    Launch LP;
    if (int e = pick_launch(rm::k_render_parked<G>, *scene, false, tune_block(), &LP)) return e;
    int gp = 2 * cu_count();                       // 2 blocks per CU: dense items, ~100 steps each
    rm::k_render_parked<G><<<gp, LP.block, LP.lds, (hipStream_t)stream>>>(a);
    return launched("k_render_parked");
  }
  return RM_OK;
}

int64_t rm_render_traj_floats(int32_t num_cameras, int32_t rows, int32_t width, int32_t steps, int32_t flags) {
  if (steps <= 0) return 0;
  return (int64_t)steps * 3 * 64 * wave_tile_count(num_cameras, rows, width, flags);
}

int64_t rm_park_floats(int64_t capacity) {
  if (capacity <= 0) return 0;
  const int64_t seg = capacity / (RM_PARK_LISTS * RM_PARK_SHARDS);
  return seg * RM_PARK_LISTS * RM_PARK_SHARDS * 4;       // ray index + 3 floats per slot
}

int rm_minmax_init(uint32_t* minmax, void* stream) {
  if (!minmax) return fail(RM_E_BADARG, "rm_minmax_init: null");
  rm::k_minmax_init<<<1, 256, 0, (hipStream_t)stream>>>(minmax);
  return launched("k_minmax_init");
}

int rm_minmax_init_many(uint32_t* minmax, int32_t count, void* stream) {
  if (!minmax || count <= 0 || count > 65535) return fail(RM_E_BADARG, "rm_minmax_init_many: bad args");
  rm::k_minmax_init<<<count, 256, 0, (hipStream_t)stream>>>(minmax);
  return launched("k_minmax_init");
}

int rm_minmax_decode(const uint32_t* minmax, float* lohi, void* stream) {
  if (!minmax || !lohi) return fail(RM_E_BADARG, "rm_minmax_decode: null");
  rm::k_minmax_decode<<<1, 64, 0, (hipStream_t)stream>>>(minmax, lohi);
  return launched("k_minmax_decode");
}

int rm_minmax_encode(const float* lohi, uint32_t* minmax, void* stream) {
  if (!minmax || !lohi) return fail(RM_E_BADARG, "rm_minmax_encode: null");
  rm::k_minmax_encode<<<1, 64, 0, (hipStream_t)stream>>>(lohi, minmax);
  return launched("k_minmax_encode");
}

int rm_shade_finish(const float* first_pass, void* image, int32_t image_dtype, int64_t n_pixels, const uint32_t* minmax,
                    int32_t mode, int32_t round_dtype, void* stream) {
  if (!first_pass || !image || !minmax || n_pixels < 0) return fail(RM_E_BADARG, "rm_shade_finish: bad args");
  if (!(io_dtype_ok(image_dtype) || image_dtype == RM_DTYPE_RGBA_F32) || !io_dtype_ok(round_dtype) ||
      (image_dtype != RM_DTYPE_F32 && (const void*)first_pass == (const void*)image))
    return fail(RM_E_BADARG, "rm_shade_finish: image dtype %d (in place only for F32)", image_dtype);
  if (!(mode == RM_MODE_DISTANCE || mode == RM_MODE_PROXIMITY || mode == RM_MODE_LAPLACIAN))
    return fail(RM_E_BADARG, "rm_shade_finish: mode %d has no second pass", mode);
  if (n_pixels == 0) return RM_OK;
  int grid = grid_for((n_pixels + 255) / 256, kMaxBlocks);
  rm::k_shade_finish<<<grid, 256, 0, (hipStream_t)stream>>>(first_pass, image, image_dtype, n_pixels, minmax, mode, round_dtype);
  return launched("k_shade_finish");
}

int rm_shade_forward(const float* px_coords, const float* orientation, const float* frames, const float* dirs,
                     const float* coords, const float* normals, const float* lap, const float* dist, void* image,
                     int32_t image_dtype, uint32_t* minmax, const void* cmap, int32_t cmap_size, int32_t cmap_dtype,
                     int32_t mode, int32_t degree, int64_t n_pixels, int64_t pixels_per_camera, void* stream) {
  if (mode < 0 || mode > 7) return fail(RM_E_BADARG, "rm_shade_forward: mode %d not in 0..7", mode);
  if (!image || n_pixels < 0 || pixels_per_camera <= 0) return fail(RM_E_BADARG, "rm_shade_forward: bad args");
  const bool need_n = (mode == RM_MODE_LAMBERTIAN || mode == RM_MODE_NORMAL || mode == RM_MODE_TANGENT || mode == RM_MODE_SPIN);
  const bool need_v = (mode == RM_MODE_LAMBERTIAN || mode == RM_MODE_VIGNETTE || mode == RM_MODE_TANGENT);
  const bool need_q = (mode == RM_MODE_TANGENT || mode == RM_MODE_SPIN);
  const bool global = (mode == RM_MODE_DISTANCE || mode == RM_MODE_PROXIMITY || mode == RM_MODE_LAPLACIAN);
  if ((need_n && !normals) || (need_v && !dirs) || (need_q && (!orientation || !cmap || cmap_size <= 0)) ||
      (mode == RM_MODE_DISTANCE && (!px_coords || !coords)) || (mode == RM_MODE_PROXIMITY && !dist) ||
      (mode == RM_MODE_LAPLACIAN && !lap) || (mode == RM_MODE_VIGNETTE && !frames) || (global && !minmax))
    return fail(RM_E_BADARG, "rm_shade_forward: an input required by mode %d is null", mode);
  if (need_q && (cmap_dtype < RM_DTYPE_F32 || cmap_dtype > RM_DTYPE_F64)) return fail(RM_E_BADARG, "rm_shade_forward: colormap dtype %d", cmap_dtype);
  if (global ? image_dtype != RM_DTYPE_F32 : !(io_dtype_ok(image_dtype) || (image_dtype == RM_DTYPE_F64 && need_q)))
    return fail(RM_E_BADARG, "rm_shade_forward: image dtype %d not valid for mode %d", image_dtype, mode);
  if (n_pixels == 0) return RM_OK;
  rm::ShadeArgs a{px_coords, orientation, frames, dirs, coords, normals, lap, dist, image, minmax, cmap,
                  cmap_size, mode, degree, n_pixels, pixels_per_camera, image_dtype, cmap_dtype};
  int grid = grid_for((n_pixels + 255) / 256, kMaxBlocks);
  rm::k_shade_fwd<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return launched("k_shade_fwd");
}

int rm_shade_backward(const float* dirs, const float* normals, const float* frames, const float* grad_image,
                      float* grad_dirs, float* grad_normals, int32_t mode, int64_t n_pixels,
                      int64_t pixels_per_camera, void* stream) {
  if (!(mode == RM_MODE_LAMBERTIAN || mode == RM_MODE_VIGNETTE || mode == RM_MODE_NORMAL))
    return fail(RM_E_BADARG, "rm_shade_backward: mode %d has no VJP (modes 0, 3, 4 do)", mode);
  if (!grad_image || n_pixels < 0 || pixels_per_camera <= 0) return fail(RM_E_BADARG, "rm_shade_backward: bad args");
  if ((mode != RM_MODE_NORMAL && !dirs) || (mode != RM_MODE_VIGNETTE && !normals) || (mode == RM_MODE_VIGNETTE && !frames))
    return fail(RM_E_BADARG, "rm_shade_backward: an input required by mode %d is null", mode);
  if (n_pixels == 0) return RM_OK;
  rm::ShadeBwdArgs a{dirs, normals, frames, grad_image, grad_dirs, grad_normals, mode, n_pixels, pixels_per_camera};
  int grid = grid_for((n_pixels + 255) / 256, kMaxBlocks);
  rm::k_shade_bwd<<<grid, 256, 0, (hipStream_t)stream>>>(a);
  return launched("k_shade_bwd");
}

int rm_shade_norm_backward(const float* raw, const float* grad_image, const float* lohi, int32_t mode, float* grad_raw,
                           float* partials, int64_t n_pixels, void* stream) {
  if (mode != RM_MODE_DISTANCE && mode != RM_MODE_PROXIMITY && mode != RM_MODE_LAPLACIAN)
    return fail(RM_E_BADARG, "rm_shade_norm_backward: mode %d is not one of the globally normalised shaders (1, 2, 5)", mode);
  if (n_pixels < 0 || (n_pixels > 0 && (!raw || !grad_image || !lohi || !grad_raw || !partials)))
    return fail(RM_E_BADARG, "rm_shade_norm_backward: null buffer");
  if (n_pixels == 0) return RM_OK;
  if (reinterpret_cast<uintptr_t>(partials) & 15) return fail(RM_E_BADARG, "rm_shade_norm_backward: partials must be 16-byte aligned");
  const int blocks = grid_for((n_pixels + 255) / 256, RM_NORM_BWD_BLOCKS);
  rm::k_shade_norm_bwd_a<<<blocks, 256, 0, (hipStream_t)stream>>>(raw, grad_image, lohi, mode, grad_raw, partials, n_pixels);
  if (int e = launched("k_shade_norm_bwd_a")) return e;
  rm::k_shade_norm_bwd_b<<<grid_for((n_pixels + 255) / 256, kMaxBlocks), 256, 0, (hipStream_t)stream>>>(raw, lohi, mode, grad_raw, partials,
                                                                                                     blocks, n_pixels);
  return launched("k_shade_norm_bwd_b");
}

int rm_camera_backward(const RmCamera* cam, const float* orientation, const float* grad_pos, const float* grad_dirs,
                       float* grad_orientation, float* grad_translation, float* partials, int32_t row_begin,
                       int32_t row_end, void* stream) {
  if (!cam || !cam->ray_positions || !cam->ray_directions || !orientation || !partials || (!grad_pos && !grad_dirs))
    return fail(RM_E_BADARG, "rm_camera_backward: null buffer");
  if (cam->dtype != RM_DTYPE_F32) return fail(RM_E_BADARG, "rm_camera_backward: fp32 camera buffers only");
  if (cam->num_cameras <= 0 || cam->num_cameras > 1024 || row_begin < 0 || row_end > cam->height || row_begin >= row_end)
    return fail(RM_E_BADARG, "rm_camera_backward: bad camera shape / rows");
  const int bpc = RM_CAMERA_BWD_BLOCKS;
  rm::k_camera_bwd<<<cam->num_cameras * bpc, 256, 0, (hipStream_t)stream>>>(*cam, orientation, grad_pos, grad_dirs, partials,
                                                                            row_begin, row_end, bpc);
  if (int e = launched("k_camera_bwd")) return e;
  rm::k_camera_bwd_finish<<<cam->num_cameras, 256, 0, (hipStream_t)stream>>>(partials, bpc, cam->num_cameras,
                                                                            grad_orientation, grad_translation);
  return launched("k_camera_bwd_finish");
}

int rm_render_backward(const RmScene* scene, const RmCamera* cam, const RmTetra* tetra, const float* orientation,
                       const float* translation, const float* traj, const int32_t* nexec, const float* p_final,
                       const float* normal_u, const float* grad_image, float* grad_params, float* partials, uint32_t* work,
                       float* grad_pos, float* grad_dirs, float* grad_qdir, const void* cmap, int32_t cmap_size,
                       int32_t cmap_dtype, int32_t mode, int32_t degree,
                       int32_t steps, int32_t row_begin, int32_t row_end, int32_t flags, int32_t* tile_cost,
                       float* hard_ws, int64_t hard_capacity, void* stream) {
#ifdef RM_NO_BACKWARD
  return fail(RM_E_BADARG, "rm_render_backward: this specialised library was built forward-only");
#else
  if (int e = check_render(scene, cam, tetra, orientation, translation, steps, row_begin, row_end)) return e;
  if (cam->dtype != RM_DTYPE_F32) return fail(RM_E_BADARG, "rm_render_backward: fp32 camera buffers only");
  const bool mapped = (mode == RM_MODE_TANGENT || mode == RM_MODE_SPIN);
  if (mode < 0 || mode > 7) return fail(RM_E_BADARG, "rm_render_backward: mode %d not in 0..7", mode);
  const int kind = (mode == RM_MODE_LAPLACIAN) ? 1 : (mode == RM_MODE_PROXIMITY ? 2 : (mode == RM_MODE_DISTANCE ? 3 : 0));
  if (mapped && (!cmap || cmap_size <= 0 || cmap_dtype < RM_DTYPE_F32 || cmap_dtype > RM_DTYPE_F64))
    return fail(RM_E_BADARG, "rm_render_backward: mode %d needs the colormap of the forward call", mode);
  if (!p_final || !grad_image || !partials || (steps > 0 && !traj)) return fail(RM_E_BADARG, "rm_render_backward: null buffer");
  rm::RenderArgs a;
  memset(&a, 0, sizeof(a));
  a.scene = *scene; a.cam = *cam; a.tetra = *tetra;
  a.orientation = orientation; a.translation = translation;
  a.traj = const_cast<float*>(traj); a.nexec = const_cast<int32_t*>(nexec); a.p_final = const_cast<float*>(p_final);
  a.normal_u = const_cast<float*>(normal_u);
  a.grad_image = grad_image; a.partials = partials; a.minmax = work;
  a.grad_pos = grad_pos; a.grad_dirs = grad_dirs; a.grad_qdir = grad_qdir; a.tile_cost = tile_cost;
  a.cmap = cmap; a.cmap_size = cmap_size; a.cmap_dtype = cmap_dtype; a.degree = degree;
  a.mode = mode; a.steps = steps; a.row_begin = row_begin; a.row_end = row_end; a.flags = flags & (RM_FLAG_TILE8X8 | RM_FLAG_DYNAMIC_TILES | RM_FLAG_EARLY_OUT);
  Launch L;
  if (int e = kind == 1 ? pick_launch(rm::k_render_bwd<GB, 1>, *scene, true, 128, &L)
            : kind == 2 ? pick_launch(rm::k_render_bwd<GB, 2>, *scene, true, 128, &L)
            : kind == 3 ? pick_launch(rm::k_render_bwd<GB, 3>, *scene, true, 128, &L)
                        : pick_launch(rm::k_render_bwd<GB, 0>, *scene, true, 128, &L)) return e;
  int64_t wave_tiles;
  {
    const int W = cam->width, rows = row_end - row_begin;
    wave_tiles = (flags & RM_FLAG_TILE8X8) ? (int64_t)cam->num_cameras * ((W + 7) >> 3) * ((rows + 7) >> 3)
                                           : ((int64_t)cam->num_cameras * rows * W + 63) / 64;
  }
  int grid = grid_for((wave_tiles + (L.block >> 6) - 1) / (L.block >> 6), tune_bwd_blocks());
  // Two tiles per wave or fewer (config 4 at 512^2: 4096 tiles, 2048 waves): the tile queues have nothing to balance, and
  // every wave would end on a look at all of them -- static stride instead (same-box: 0.340 -> 0.329 ms per step at 512^2;
  // at 1024^2, 8 tiles per wave, the queues win by 2 %: profiles/r03_train_ab.txt)
  if (wave_tiles <= 2 * (int64_t)grid * (L.block >> 6) && !env_set("RM_BWD_DYNAMIC_TILES")) a.flags &= ~RM_FLAG_DYNAMIC_TILES;
  // deferred rays: only with the reverse early exit, a workspace for the list and its counter
  const bool defer = hard_ws && hard_capacity > 0 && work && (flags & RM_FLAG_EARLY_OUT) && steps > 0;
  if (defer) {
    if (hard_capacity > (int64_t)1 << 21 || steps >= 2048)
      return fail(RM_E_BADARG, "rm_render_backward: deferred-ray list limited to 2^21 rays and 2047 steps");
    const int64_t cap = hard_capacity;
    // float4 accesses of hard_n / hard_p: 16-byte alignment of the workspace is the caller's (a fresh device allocation);
    // the arrays in front of them are kept to multiples of 4 floats by rounding the capacity down
    a.hard_cap = (int32_t)(cap & ~(int64_t)3);
    const int64_t c4 = a.hard_cap;
    if (c4 < 4 || (reinterpret_cast<uintptr_t>(hard_ws) & 15)) return fail(RM_E_BADARG, "rm_render_backward: hard_ws must be 16-byte aligned, capacity >= 4");
    a.hard_n = hard_ws;                                                   // [steps][c4][4]
    a.hard_p = hard_ws + (int64_t)steps * c4 * 4;                        // [steps][c4][4]
    a.hard_state = hard_ws + (int64_t)steps * c4 * 8;                    // [c4][8]
    a.hard_g = a.hard_state + 8 * c4;                                     // [steps][c4]
    a.hard_pairs = reinterpret_cast<uint32_t*>(a.hard_g + (int64_t)steps * c4);
    a.hard_ray = reinterpret_cast<int32_t*>(a.hard_pairs + (int64_t)steps * c4);
    a.hard_slot = a.hard_ray + c4;
    a.hard_step = a.hard_slot + c4;
  }
  if (kind == 1) rm::k_render_bwd<GB, 1><<<grid, L.block, L.lds, (hipStream_t)stream>>>(a);
  else if (kind == 2) rm::k_render_bwd<GB, 2><<<grid, L.block, L.lds, (hipStream_t)stream>>>(a);
  else if (kind == 3) rm::k_render_bwd<GB, 3><<<grid, L.block, L.lds, (hipStream_t)stream>>>(a);
  else rm::k_render_bwd<GB, 0><<<grid, L.block, L.lds, (hipStream_t)stream>>>(a);
  if (int e = launched("k_render_bwd")) return e;
  int rows = grid;
  if (defer) {
    Launch LN, LB;
    if (int e = pick_launch(rm::k_bwd_hard_n<GB>, *scene, true, 128, &LN)) return e;
    if (int e = pick_launch(rm::k_bwd_hard_b<GB>, *scene, true, 128, &LB)) return e;
    rm::k_bwd_hard_n<GB><<<tune_hardn_blocks(), LN.block, LN.lds, (hipStream_t)stream>>>(a);   // no accumulators: 4 waves / SIMD
    if (int e = launched("k_bwd_hard_n")) return e;
    rm::k_bwd_hard_a<<<(int)((a.hard_cap + 63) / 64), 64, 0, (hipStream_t)stream>>>(a);   // one wave per block: more CUs busy
    if (int e = launched("k_bwd_hard_a")) return e;
    rm::RenderArgs b = a;
    b.partials = partials + (size_t)rows * (scene->n_params + scene->n_grad_derived);     // its rows follow k_render_bwd's
    rm::k_bwd_hard_b<GB><<<tune_hardb_blocks(), LB.block, LB.lds, (hipStream_t)stream>>>(b);
    if (int e = launched("k_bwd_hard_b")) return e;
    rows += tune_hardb_blocks();
  }
  return reduce_partials(*scene, partials, rows, grad_params, (hipStream_t)stream);
#endif
}

int rm_tile_order_from_cost(const int32_t* tile_cost, int64_t n_tiles, int32_t max_cost, int32_t* tile_order,
                            int32_t* scratch, void* stream) {
  if (!tile_cost || !tile_order || n_tiles <= 0 || n_tiles > 0x7fffffff || max_cost < 0)
    return fail(RM_E_BADARG, "rm_tile_order_from_cost: bad args");
  if (n_tiles > RM_ORDER_ONE_BLOCK && !scratch)
    return fail(RM_E_BADARG, "rm_tile_order_from_cost: more than %d items need the scratch buffer", RM_ORDER_ONE_BLOCK);
  const size_t lds = (32 * 1024 + 64) * sizeof(int);      // 128 KiB of gfx950's 160 KiB
  static std::atomic<bool> attr_set[RM_MAX_DEVICES];       // per device: function attributes do not carry over
  const int slot = current_device_slot();
  if (slot < 0 || !attr_set[slot].load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rm::k_order_scatter),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(RM_E_LAUNCH, "hipFuncSetAttribute(k_order_scatter): %s", hipGetErrorString(e));
    if (slot >= 0) attr_set[slot].store(true, std::memory_order_release);
  }
  // one block up to 4096 items (or without scratch), else ~4096 items per block: 1080p tiles 8 blocks, rays 256
  int blocks = 1;
  if (scratch && n_tiles > 4096) {
    blocks = (int)((n_tiles + 4095) / 4096);
    if (blocks > RM_ORDER_SCRATCH_INTS / 32) blocks = RM_ORDER_SCRATCH_INTS / 32;
  }
  if (blocks > 1) {
    rm::k_order_count<<<blocks, 1024, 0, (hipStream_t)stream>>>(tile_cost, n_tiles, max_cost, scratch);
    if (int e = launched("k_order_count")) return e;
  }
  rm::k_order_scatter<<<blocks, 1024, lds, (hipStream_t)stream>>>(tile_cost, n_tiles, max_cost, scratch, tile_order);
  return launched("k_order_scatter");
}

int rm_tile_score_from_ray_cost(const int32_t* ray_cost, int64_t n_tiles, int32_t tiles_x, int32_t tiles_y, int32_t reach,
                                int32_t max_cost, int32_t* raw, int32_t* tile_score, void* stream) {
  if (!ray_cost || !tile_score || !raw || n_tiles <= 0 || n_tiles > 0x1ffffff || max_cost < 0 || tiles_x <= 0 || tiles_y <= 0 ||
      n_tiles % ((int64_t)tiles_x * tiles_y) != 0 || reach < 0 || reach > 8)
    return fail(RM_E_BADARG, "rm_tile_score_from_ray_cost: bad args");
  rm::k_tile_score_raw<<<grid_for((n_tiles + 3) / 4, kMaxBlocks), 256, 0, (hipStream_t)stream>>>(ray_cost, n_tiles, max_cost, raw);
  if (int e = launched("k_tile_score_raw")) return e;
  rm::k_tile_score_classes<<<grid_for((n_tiles + 255) / 256, kMaxBlocks), 256, 0, (hipStream_t)stream>>>(
      raw, n_tiles, tiles_x, tiles_y, reach, max_cost, tile_score);
  return launched("k_tile_score_classes");
}

int rm_sum_rows(const float* rows, int64_t n_rows, int32_t width, float* out, void* stream) {
  if (!rows || !out || n_rows < 0 || n_rows > 0x7fffffff || width <= 0 || width > 65535)
    return fail(RM_E_BADARG, "rm_sum_rows: bad args");
  rm::k_reduce_partials<<<width, 256, 0, (hipStream_t)stream>>>(rows, (int)n_rows, width, out);
  return launched("k_reduce_partials");
}

/* Host-side validation of a compiled program (host pointer).  The device copy a
 * RmScene points at cannot be inspected without a synchronising copy, so the host
 * compiler validates once here before uploading. */
int rm_validate_program(const int32_t* host_program, int32_t n_instr, int32_t n_params, int32_t n_derived,
                        int32_t stack_floats, int32_t n_slots) {
  if (!host_program || n_instr <= 0) return fail(RM_E_PROGRAM, "empty program");
  static const int psize[RM_OP__COUNT] = {0, 1, 3, 0, 7, 1, 2, 7, 7, 0, 0, 0, 0, 1, 1, 1, 1, 0, 0};
  int depth_f = 0, depth_b = 0, max_f = 0, max_b = 0, values = 0;
  for (int i = 0; i < n_instr; ++i) {
    const int32_t* w = host_program + 4 * i;
    int op = w[0], off = w[1], a0 = w[2], a1 = w[3];
    if (op <= RM_OP_END || op >= RM_OP__COUNT) return fail(RM_E_PROGRAM, "instr %d: bad opcode %d", i, op);
    if (psize[op] && (off < 0 || off + psize[op] > n_params)) return fail(RM_E_PROGRAM, "instr %d: params out of range", i);
    switch (op) {
      case RM_OP_LINE:
        if (a0 < n_params || a0 + 6 > n_params + n_derived) return fail(RM_E_PROGRAM, "instr %d: derived block out of range", i);
        values++;
        break;
      case RM_OP_SPHERE: case RM_OP_BOX: case RM_OP_PLANE: case RM_OP_DISK: case RM_OP_TORUS:
        values++;
        break;
      case RM_OP_AFFINE_PUSH: depth_f += 3; depth_b += 6; break;
      case RM_OP_AFFINE_POP: depth_f -= 3; depth_b -= 6; break;
      case RM_OP_UNION_BEGIN: depth_f += 1; depth_b += 2; break;
      case RM_OP_SMOOTH_BEGIN:
        depth_b += 2;
        if (a0 != 0) {     // bound table of a smooth union (for the culling of its children and / or of the union as a whole)
          const int base = (a1 >> 8) & 255, n = a1 & 255;
          if ((a1 >> 16) & ~1) return fail(RM_E_PROGRAM, "instr %d: SMOOTH_BEGIN flags", i);
          if (off < 0 || off >= n_params) return fail(RM_E_PROGRAM, "instr %d: SMOOTH_BEGIN blend_k out of range", i);
          if (n < 1 || n > 64 || base < 0 || base + n > 64 || base + n > n_slots || a0 < n_params || (a0 & 3) ||
              a0 + 8 * n > n_params + n_derived)
            return fail(RM_E_PROGRAM, "instr %d: SMOOTH_BEGIN bound table out of range", i);
        }
        break;
      case RM_OP_UNION_END: case RM_OP_SMOOTH_END:
        if (a1 <= 0 || a0 < 0 || a0 + a1 + (op == RM_OP_SMOOTH_END ? 1 : 0) > n_slots)
          return fail(RM_E_PROGRAM, "instr %d: slots out of range", i);
        if (op == RM_OP_SMOOTH_END && a1 >= 512) return fail(RM_E_PROGRAM, "instr %d: smooth union of %d >= 512 children", i, a1);
        depth_f -= (op == RM_OP_UNION_END) ? 1 : 0; depth_b -= 2;
        values++;
        break;
      case RM_OP_FOLD_MIN: case RM_OP_FOLD_LSE: case RM_OP_ONION:
        if (a0 < 0 || a0 >= n_slots) return fail(RM_E_PROGRAM, "instr %d: slot out of range", i);
        if (op != RM_OP_ONION) values--;
        if (op == RM_OP_FOLD_LSE && a1 != 0) {
          const int c = i - a1;
          if (a1 < 0 || c < 0 || host_program[4 * c] != RM_OP_CULL_LSE || host_program[4 * c + 2] != a0 || host_program[4 * c + 3] != a1)
            return fail(RM_E_PROGRAM, "instr %d: FOLD_LSE does not point back at its CULL_LSE", i);
        }
        if (op == RM_OP_FOLD_MIN && a1 != 0) {
          const int c = i - a1;
          if (a1 < 0 || c < 0 || host_program[4 * c] != RM_OP_CULL_MIN || host_program[4 * c + 3] != ((a1 << 8) | a0))
            return fail(RM_E_PROGRAM, "instr %d: FOLD_MIN does not point back at its CULL_MIN", i);
        }
        break;
      case RM_OP_CULL_LSE:
        if (off < n_params || (off & 3) || off + 8 > n_params + n_derived) return fail(RM_E_PROGRAM, "instr %d: bound entry out of range", i);
        if (a1 < 2 || i + a1 >= n_instr || a0 < 0 || a0 >= 64 || a0 >= n_slots || host_program[4 * (i + a1)] != RM_OP_FOLD_LSE ||
            host_program[4 * (i + a1) + 2] != a0 || host_program[4 * (i + a1) + 3] != a1)
          return fail(RM_E_PROGRAM, "instr %d: CULL_LSE does not match its FOLD_LSE", i);
        break;
      case RM_OP_CULL_MIN: {
        const int skip = a1 >> 8, slot = a1 & 255;
        if (a0 < n_params || a0 + 5 > n_params + n_derived) return fail(RM_E_PROGRAM, "instr %d: bound out of range", i);
        if (off != 0 && (off != 1 || i + 1 >= n_instr || host_program[4 * (i + 1)] != RM_OP_SMOOTH_BEGIN || host_program[4 * (i + 1) + 2] == 0))
          return fail(RM_E_PROGRAM, "instr %d: CULL_MIN by children needs a SMOOTH_BEGIN with a bound table next", i);
        if (skip < 2 || i + skip >= n_instr || slot >= 64 || slot >= n_slots ||
            host_program[4 * (i + skip)] != RM_OP_FOLD_MIN || host_program[4 * (i + skip) + 2] != slot ||
            host_program[4 * (i + skip) + 3] != skip)
          return fail(RM_E_PROGRAM, "instr %d: CULL_MIN does not match its FOLD_MIN", i);
      } break;
      default: break;
    }
    if (depth_f < 0 || depth_b < 0 || values < 0 || values > 1)
      return fail(RM_E_PROGRAM, "instr %d: unbalanced program (value register holds %d values)", i, values);
    if (depth_f > max_f) max_f = depth_f;
    if (depth_b > max_b) max_b = depth_b;
  }
  if (depth_f != 0 || depth_b != 0) return fail(RM_E_PROGRAM, "unbalanced program at end");
  if (values != 1) return fail(RM_E_PROGRAM, "program leaves %d values (expected 1)", values);
  if (max_b > stack_floats || max_f > stack_floats) return fail(RM_E_PROGRAM, "stack_floats %d < needed %d", stack_floats, max_b);
  return RM_OK;
}

}  // extern "C"

/*
 * rm_abi.h -- C ABI of the MI355X (gfx950) sphere-tracing library, librm_hip.so.
 *
 * The reference (kyle-rosa/ray_marching) has no FFI boundary: its operator API
 * is Python nn.Module.__call__ on torch tensors.  These entry points are what a
 * binding for its hot path binds instead of the ATen op stream; each one cites
 * the reference interface it replaces.  Plain pointers and sizes only, no torch
 * types.  The Python host (ray_marching_amd/_abi.py) calls them through ctypes
 * with tensor.data_ptr() and the current HIP stream handle.
 *
 * Conventions
 *   - every pointer marked "device" is HBM memory owned by the caller; the
 *     library never allocates, frees or synchronises;
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream); calls are re-entrant per stream;
 *   - return value 0 = enqueued, negative = error (RM_E_*), message available
 *     from rm_last_error() on the calling thread; no C++ exception crosses;
 *   - all arithmetic is IEEE fp32.  Arrays are contiguous, channels-last: points [n,3], images
 *     [N,rows,W,3].  The forward entry points take the element type of their I/O arrays as an
 *     RM_DTYPE_* argument (the reference module cast with .to(dtype), main.py:20-26 runs float16):
 *     fp16 arrays are converted in the kernels' loads and stores, there is no separate cast pass.
 *     Backward entry points are fp32 only.
 */
#ifndef RM_ABI_H
#define RM_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RM_ABI_VERSION 13

enum {
  RM_DTYPE_F32 = 0,
  RM_DTYPE_F16 = 1,      /* IEEE binary16 storage, fp32 arithmetic */
  RM_DTYPE_F64 = 2,      /* images / colormap of the tangent and spin shaders only: the reference multiplies an
                            fp32 brightness by its float64 colormap, giving a float64 image (shader.py:104,118) */
  RM_DTYPE_RGBA_F32 = 3  /* image of rm_render_forward / rm_shade_finish only, one camera: [rows,W,4] fp32 = the display
                            contract of main.py:78-84, F.pad(images.mean(0).float(), [0,1], 1.0) (torchwindow/window.py:146-174:
                            contiguous RGBA32F, row pitch 16 W) written by the frame kernel itself -- the pixel rounded to the
                            type the image would have had (cam->dtype; float64 for modes 6, 7 with a float64 colormap), then
                            to fp32, alpha = 1 */
};

enum {
  RM_OK = 0,
  RM_E_BADARG = -1,      /* null pointer / negative size / bad mode */
  RM_E_PROGRAM = -2,     /* scene program failed validation */
  RM_E_TOO_LARGE = -3,   /* scene does not fit the LDS budget */
  RM_E_LAUNCH = -4       /* HIP launch error */
};

/* ---- scene program ------------------------------------------------------
 * A scene (the reference's nn.Module tree: scene/primitives.py:6-102,
 * scene/transformations.py:8-132) is compiled by the host into a flat list of
 * 4-word instructions {opcode, param_offset, aux0, aux1} evaluated left to
 * right, plus one packed fp32 parameter block in named_parameters() order.
 */
enum {
  RM_OP_END = 0,
  RM_OP_SPHERE = 1,        /* P: radius                     primitives.py:11-17  */
  RM_OP_BOX = 2,           /* P: halfsides[3]               primitives.py:25-33  */
  RM_OP_PLANE = 3,         /*                               primitives.py:40-41  */
  RM_OP_LINE = 4,          /* P: start[3] end[3] radius; aux0 = derived offset (AB[3], AB/|AB|^2[3])  primitives.py:51-61 */
  RM_OP_DISK = 5,          /* P: radius                     primitives.py:69-82  */
  RM_OP_TORUS = 6,         /* P: radius1 radius2            primitives.py:91-102 */
  RM_OP_AFFINE_PUSH = 7,   /* P: translation[3] orientation[4]   transformations.py:33-42 */
  RM_OP_AFFINE_POP = 8,    /* P: same offset as the matching PUSH */
  RM_OP_UNION_BEGIN = 9,   /*                               transformations.py:90-94 */
  RM_OP_FOLD_MIN = 10,     /* aux0 = tape slot; aux1 = distance back to its CULL_MIN (0 = none) */
  RM_OP_UNION_END = 11,    /* aux0 = first tape slot, aux1 = child count */
  RM_OP_SMOOTH_BEGIN = 12, /* P: blend_k (read only when aux0 != 0); aux0 = derived offset of the children's bound table, 8 floats per
                              child in slot order, 16-byte aligned, written by the kernels at staging time (0 = none); aux1 =
                              (children are culled: RM_OP_CULL_LSE) << 16 | first tape slot << 8 | child count (<= 64 children,
                              slots < 64)    transformations.py:67-71 */
  RM_OP_FOLD_LSE = 13,     /* P: blend_k; aux0 = tape slot; aux1 = distance back to its CULL_LSE (0 = none) */
  RM_OP_SMOOTH_END = 14,   /* P: blend_k; aux0 = first tape slot, aux1 = child count; slot aux0+aux1 holds the logsumexp */
  RM_OP_ROUND = 15,        /* P: rounding                   transformations.py:117-118 */
  RM_OP_ONION = 16,        /* P: radius; aux0 = tape slot   transformations.py:131-132 */
  RM_OP_CULL_MIN = 17,     /* before a child of an SDFUnion: `param offset` field = 1 when the child is a smooth union whose SMOOTH_BEGIN
                              (the next instruction) carries a bound table: the union is then also skipped when the minimum of
                              its children's own lower bounds minus log(n) / blend_k cannot lower the running minimum, else 0;
                              aux0 = derived offset of its bound, 5 floats
                              {cx,cy,cz,K,slope} written by the kernels at staging time; aux1 = (n << 8) | slot, n = instructions up to and including the
                              child's FOLD_MIN (whose aux1 = n).  Skips the child when it cannot lower the
                              running minimum for any ray of the wave (exact) */
  RM_OP_CULL_LSE = 18,     /* before a child of an SDFSmoothUnion whose SMOOTH_BEGIN carries a bound table: `param offset` field =
                              derived offset of this child's table entry {cx,cy,cz,slope_lb,K_lb,slope_ub,K_ub,0} (written by the
                              kernels at staging time), aux0 = the child's tape slot, aux1 = n = instructions up to and including
                              the child's FOLD_LSE (whose aux1 = n).  Skips the child when its term of the logsumexp is exactly
                              +0.0f for every ray of the wave: k (d_i - d_min) > 104 (exact) */
  RM_OP__COUNT = 19
};

/* Where one float of the parameter block lives: element `elem` of a device array of `dtype` (F32 or F16).
 * A table of n_params of these lets the kernels gather the block straight from the nn.Parameter storages in
 * their prologue, so an inference frame needs no packing pass and can never see stale values (in-place edits,
 * optimiser steps and .data writes are all just memory the next launch reads). */
typedef struct RmParamRef {
  const void* base;
  int32_t elem;
  int32_t dtype;
} RmParamRef;

typedef struct RmScene {
  const int32_t* program;  /* device, n_instr * 4 int32 */
  const float* params;     /* device, n_params fp32 (raw parameters, packed); may be NULL when param_refs is given */
  const RmParamRef* param_refs; /* device, n_params entries, or NULL: gather the block from here instead of `params` */
  int32_t n_instr;
  int32_t n_params;        /* raw parameter floats = length of every grad_params vector */
  int32_t n_derived;       /* derived constants appended in LDS after the raw block */
  int32_t stack_floats;    /* per-ray evaluation stack depth (floats) */
  int32_t n_slots;         /* per-ray tape slots (fold / onion inputs) */
  int32_t n_grad_derived;  /* leading floats of the derived block that carry gradients (capsule constants); the rest
                              (cull bounds, bound tables) has none, so the backward kernels keep
                              n_params + n_grad_derived accumulators per ray */
  const float* block;      /* device, n_params + n_derived fp32, or NULL: the FINISHED scene block (raw parameters, then the
                              derived constants) as a previous launch left it in `block_out`.  Every block of a launch gathers
                              the parameters and derives the constants (capsule axes, bounding spheres of the cull tests, bound
                              tables) in its prologue -- 9-20 us for the closed 6-primitive scene, 120 us for the 32-primitive
                              one, one thread walking the program -- unless it is handed the result here.  The kernels of a
                              backward pass take the block of their forward pass (same parameters by construction: autograd) */
  float* block_out;        /* device, n_params + n_derived fp32, or NULL: block 0 of the launch writes its staged block here */
  float* block_cache;      /* device, n_params + n_derived fp32, or NULL: a finished block the launch may REUSE AFTER CHECKING IT.
                              Every block gathers the live parameters as always and compares them, bit for bit, with the
                              parameters the cache was derived from; equal: it takes the cached derived constants (the walk over
                              the program is skipped: 2-5 us per block of the reference's scenes); different, or a cache the caller
                              filled with 0xFFFFFFFF words: it derives them, and block 0 rewrites the cache (derived constants
                              first, then -- behind a device-scope fence -- the parameters they belong to; readers load with
                              device scope).  In-place edits, optimiser steps and .data writes are therefore seen by the very next
                              launch, as without it.  One cache per stream: launches that may run concurrently must not share
                              one.  Ignored when `block` is given or n_params == 0 */
} RmScene;

/* PinholeCamera buffers (rendering/ray_marching.py:26-50). */
typedef struct RmCamera {
  const void* ray_positions;    /* device [N,H,W,3] camera-frame origins */
  const void* ray_directions;   /* device [N,H,W,3] camera-frame unit directions */
  int32_t num_cameras, height, width;
  int32_t dtype;                /* RM_DTYPE_F32 or RM_DTYPE_F16: element type of the two buffers and of the
                                   orientation / translation arrays passed along with them */
} RmCamera;

/* SDFNormals constants (rendering/ray_marching.py:96-113), host memory. */
typedef struct RmTetra {
  float offsets[12];   /* 4 taps x 3, already scaled by normals_eps */
  float inverse[9];    /* inverse of the relative offsets, row major */
  float lap_scale;     /* 6 / eps^2 */
} RmTetra;

/* shader modes, order of rendering/shader.py:204-209 */
enum {
  RM_MODE_LAMBERTIAN = 0, RM_MODE_DISTANCE = 1, RM_MODE_PROXIMITY = 2, RM_MODE_VIGNETTE = 3,
  RM_MODE_NORMAL = 4, RM_MODE_LAPLACIAN = 5, RM_MODE_TANGENT = 6, RM_MODE_SPIN = 7
};

/* flags */
enum {
  RM_FLAG_EARLY_OUT = 1,   /* wave-uniform exit once every ray of the wave is in a proven bit-exact
                              cycle of the march map (results unchanged) */
  RM_FLAG_TILE8X8 = 2,     /* a wave covers an 8x8 pixel tile instead of 64 pixels of one row */
  RM_FLAG_DYNAMIC_TILES = 4, /* waves draw tiles from the workspace's atomic tile queues instead
                              of a static stride */
  RM_FLAG_ORDER_PER_RAY = 16, /* with RM_FLAG_REGEN: `tile_order` has one entry per ray slot (below) instead of per tile */
  RM_FLAG_REGEN = 8        /* rm_render_forward only: ray regeneration.  A wave is a pool of 64 ray slots; a lane whose
                              ray has reached its final iterate is handed the next ray of a queue instead of idling
                              until the slowest ray of its tile is done.  Two launches (march into `p_final`, then
                              distance / normals / shader per tile); same image, bit for bit.  Pays where few rays
                              of a tile settle late (the reference's default pose inside the torus: 2.5 ray-steps
                              executed per step needed without it); costs a few per cent where tiles settle
                              together.  Requires RM_FLAG_EARLY_OUT, RM_FLAG_TILE8X8, `minmax`, `p_final`,
                              steps % 4 == 0, and no traj / nexec / parking.  `tile_order` matters more here than for
                              the tile kernel: once the queues are dry idle lanes cannot be refilled, so the rays
                              that march longest should be dealt first.  With this flag `tile_cost` has one entry
                              per RAY SLOT (64 * rm_wave_tiles() entries, slot = tile * 64 + lane of the 8x8 tile) and
                              receives the steps each ray needed; rm_tile_score_from_ray_cost turns it into a score
                              per tile for rm_tile_order_from_cost (robust under camera motion: WHICH rays of a tile
                              march long changes with the least move, how many do not).  RM_FLAG_ORDER_PER_RAY:
                              `tile_order` is a permutation of the ray slots instead (rm_tile_order_from_cost of the
                              per-ray costs: the best order for a frame that is rendered again unchanged). */
};

/* Workspace ("minmax") layout, uint32 words, prepared by rm_minmax_init before every launch
 * that uses it: [0] global min, [1] global max (order-preserving encoding), [2] NaN flag -- as written by
 * rm_minmax_encode -- plus RM_WORK_MM_SLOTS partial {min, max, NaN flag} triples at
 * [RM_WORK_MM_BASE + s*32], one 128-B line each, which the frame kernels fold their waves' values into (5000 waves
 * folding into ONE word triple cost 60-100 us of serialised atomics per frame); the global value is the combination
 * of all of them (rm_minmax_decode, rm_shade_finish).
 * [RM_WORK_QUEUE_BASE + q*RM_WORK_QUEUE_STRIDE] tile counter of queue q. */
#define RM_WORK_QUEUES 64
#define RM_WORK_QUEUE_STRIDE 32
#define RM_WORK_QUEUE_BASE 64
#define RM_WORK_PARK_BASE (RM_WORK_QUEUE_BASE + RM_WORK_QUEUES * RM_WORK_QUEUE_STRIDE)
#define RM_PARK_LISTS 8          /* one list of parked rays per check step (see rm_render_forward: park_ws) */
#define RM_PARK_SHARDS 4         /* counters per list, each on its own 128-B line */
#define RM_WORK_MM_BASE (RM_WORK_PARK_BASE + RM_PARK_LISTS * RM_PARK_SHARDS * 32)
#define RM_WORK_MM_SLOTS 64
#define RM_WORK_WORDS (RM_WORK_MM_BASE + RM_WORK_MM_SLOTS * 32)

int rm_abi_version(void);
const char* rm_last_error(void);

/* Workspace sizing for the backward entry points: number of floats of
 * `partials` needed for a launch over n rays. */
int64_t rm_grad_partials_floats(const RmScene* scene, int64_t n);

/* scene(query[...,3]) -> [...,1]          (every forward() in scene/primitives.py, transformations.py) */
int rm_sdf_forward(const RmScene* scene, const void* points /*device [n,3]*/,
                   void* dist /*device [n]*/, int64_t n, int32_t dtype /*F32 | F16, both arrays*/, void* stream);

/* VJP of rm_sdf_forward: grad_points[n,3] (nullable) and grad_params[n_params]
 * (nullable, OVERWRITTEN with the deterministic sum over rays). */
int rm_sdf_backward(const RmScene* scene, const float* points, const float* grad_dist /*[n]*/,
                    float* grad_points, float* grad_params, float* partials, int64_t n, void* stream);

/* SDFMarcher.forward (rendering/ray_marching.py:72-84): p <- f(p)*v + p, `steps` times.
 * traj (nullable): device [steps,n,3], iterate p_i BEFORE step i (needed by backward).
 * nexec (nullable): device int32 [n], steps executed before the early-out fixed point. */
int rm_march_forward(const RmScene* scene, const void* pos, const void* dirs, void* out_pos,
                     float* traj, int32_t* nexec, int64_t n, int32_t steps, int32_t flags,
                     int32_t dtype /*F32 | F16: pos, dirs, out_pos; traj is always fp32*/, void* stream);

/* VJP of rm_march_forward w.r.t. pos, dirs (nullable) and parameters. */
int rm_march_backward(const RmScene* scene, const float* dirs, const float* traj, const int32_t* nexec,
                      const float* grad_out /*[n,3]*/, float* grad_pos, float* grad_dirs,
                      float* grad_params, float* partials, int64_t n, int32_t steps, void* stream);

/* SDFNormals.forward (rendering/ray_marching.py:115-125). */
int rm_normals_forward(const RmScene* scene, const RmTetra* tetra, const void* coords /*[n,3]*/,
                       void* normals /*[n,3]*/, void* laplacian /*[n]*/, int64_t n,
                       int32_t dtype /*F32 | F16, all three arrays*/, void* stream);

int rm_normals_backward(const RmScene* scene, const RmTetra* tetra, const float* coords,
                        const float* grad_normals /*[n,3] nullable*/, const float* grad_lap /*[n] nullable*/,
                        float* grad_coords, float* grad_params, float* partials, int64_t n, void* stream);

/* PinholeCamera.forward (rendering/ray_marching.py:57-64). frames: [N,3,3].  Every array has cam->dtype. */
int rm_camera_forward(const RmCamera* cam, const void* orientation /*device [N,4]*/,
                      const void* translation /*device [N,3]*/, void* out_pos, void* out_dirs,
                      void* out_frames, void* stream);

/* RenderLoop.forward (control.py:231-258), fused: camera -> march -> distance ->
 * normals/laplacian -> shader, rows [row_begin,row_end) of every camera.
 *   orientation, translation : device [N,4], [N,3], element type cam->dtype.
 *   image   : device [N,rows,W,3] of image_dtype (F32, F16; F64 for modes 6,7 with a float64 colormap).
 *   p_final : nullable fp32 [N,rows,W,3]; nexec nullable int32 [N,rows,W].
 *   traj    : nullable fp32, rm_render_traj_floats(...) floats: the iterates of the march for rm_render_backward, laid
 *             out [wave tile][step][component][lane]: every store of a wave is 256 contiguous bytes and a tile's whole
 *             trajectory one contiguous block; private
 *             to this pair of entry points.
 *   normal_u : nullable fp32 [N,rows,W,3]: the un-normalised normal of the final point (the normal is normal_u /
 *             |normal_u|); handed to rm_render_backward it saves that kernel eight scene evaluations per ray.
 *   minmax  : device uint32[RM_WORK_WORDS] prepared by rm_minmax_init.  Holds the tile queues of
 *             RM_FLAG_DYNAMIC_TILES (NULL = static striding).  Words 0-2 are required for modes 1,2,5
 *             (global min/max, shader.py:35-36, 52-53, 84).
 *   first_pass : modes 1,2,5 only: device fp32 [N,rows,W,3] receiving the un-normalised value (it may alias
 *             `image` when image_dtype is F32); the kernel folds its min/max into minmax and
 *             rm_shade_finish(first_pass -> image) normalises.  Between the two calls a multi-GPU host
 *             all-reduces minmax (rm_minmax_* helpers).
 *   cmap    : device [cmap_size,3] of cmap_dtype (F32, F16 or F64), required for modes 6,7.
 *   tile_order : nullable device int32[T] permutation of the T wave tiles (T = rm_wave_tiles()): the tile
 *             dealt at position i is tile_order[i].  Waves draw positions in increasing order, so an order
 *             sorted by decreasing cost (e.g. the previous frame's tile_cost) shortens the tail of the launch;
 *             any permutation gives the same image.
 *   tile_cost : nullable device int32[T] out: march steps the wave of each tile executed.
 *   park_ws, park_capacity : nullable workspace of rm_park_floats(park_capacity) floats.  A wave leaves a tile when
 *             ALL its 64 rays have settled into a bit-exact cycle; where only a minority is still moving (the
 *             reference's default pose inside the torus: a third of the tiles run all 128 steps for 16 of their
 *             64 rays on average), those rays are parked on a list at a check step and a second kernel
 *             (k_render_parked) marches them on in dense waves and shades them.  Same pixels, bit for bit.
 *             Requires RM_FLAG_EARLY_OUT, `minmax`, no trajectory recording, and a library built with
 *             -DRM_PARKING (an opt-in: measured +12 % at that pose, -8 % on the headline frame); NULL / 0 = off.
 */
int rm_render_forward(const RmScene* scene, const RmCamera* cam, const RmTetra* tetra,
                      const void* orientation, const void* translation,
                      void* image, int32_t image_dtype, float* first_pass, float* p_final, float* traj, int32_t* nexec,
                      float* normal_u, uint32_t* minmax, const void* cmap, int32_t cmap_size, int32_t cmap_dtype,
                      int32_t mode, int32_t degree, int32_t steps,
                      int32_t row_begin, int32_t row_end, int32_t flags,
                      const int32_t* tile_order, int32_t* tile_cost,
                      float* park_ws, int64_t park_capacity, void* stream);

/* floats of the `traj` buffer of rm_render_forward / rm_render_backward for a band of `rows` rows */
int64_t rm_render_traj_floats(int32_t num_cameras, int32_t rows, int32_t width, int32_t steps, int32_t flags);

/* floats of a parking workspace for up to `capacity` rays (4 per ray: pixel index and the iterate) */
int64_t rm_park_floats(int64_t capacity);

/* number of wave tiles (64-ray work units) of a band of `rows` rows: length of tile_order / tile_cost */
int64_t rm_wave_tiles(int32_t num_cameras, int32_t rows, int32_t width, int32_t flags);

/* tile_order for the NEXT frame from this frame's tile_cost (values 0 .. max_cost = the step count): items by
 * decreasing cost class (32 classes), natural order inside a class (stable counting sort).  Worth it when consecutive
 * frames are coherent (an interactive camera).  `scratch` = device int32[RM_ORDER_SCRATCH_INTS] lets several blocks
 * share the work (1080p tiles: 96 us with one block, ~10 us with 8); NULL is allowed up to RM_ORDER_ONE_BLOCK items,
 * which one block then sorts alone. */
/* Tile scores (0..31: sort with max_cost = 31) from the per-ray costs RM_FLAG_REGEN recorded, for frames of
 * n_tiles = cameras * tiles_x * tiles_y 8x8 tiles: 31..17 tiles with rays that marched >= 3/4 of max_cost steps, by their
 * number; 16 tiles without, within `reach` tiles of one (where such rays turn up when the camera moves by up to
 * 8 * reach pixels before the order is renewed); 15..0 the rest by the longest ray in that neighbourhood.
 * raw: device scratch int32[n_tiles]. */
int rm_tile_score_from_ray_cost(const int32_t* ray_cost /*[64 * n_tiles]*/, int64_t n_tiles, int32_t tiles_x, int32_t tiles_y,
                                int32_t reach, int32_t max_cost, int32_t* raw, int32_t* tile_score /*[n_tiles]*/, void* stream);
#define RM_ORDER_ONE_BLOCK 131072
#define RM_ORDER_SCRATCH_INTS 8192
int rm_tile_order_from_cost(const int32_t* tile_cost, int64_t n_tiles, int32_t max_cost, int32_t* tile_order,
                            int32_t* scratch /*nullable*/, void* stream);

/* workspace helpers: init (min=+inf, max=-inf, no NaN, all tile counters 0; the buffer holds
 * RM_WORK_WORDS uint32); decode to two floats {lo, hi};
 * encode two floats back (after a host-side all-reduce). */
int rm_minmax_init(uint32_t* minmax /*device*/, void* stream);
/* `count` workspaces laid out back to back (count * RM_WORK_WORDS words) in ONE launch: a host that renders frame after
 * frame prepares a batch and hands one to every launch (5 us per frame otherwise, 2 % of a 1080p frame) */
int rm_minmax_init_many(uint32_t* minmax /*device*/, int32_t count, void* stream);
int rm_minmax_decode(const uint32_t* minmax, float* lohi /*device [2]*/, void* stream);
int rm_minmax_encode(const float* lohi /*device [2]*/, uint32_t* minmax, void* stream);

/* second pass of the globally normalised shaders (modes 1, 2, 5): first_pass fp32 [n_pixels,3] -> image
 * [n_pixels,3] of image_dtype (F32 or F16); in place when the two pointers are equal (F32 only). */
int rm_shade_finish(const float* first_pass, void* image, int32_t image_dtype, int64_t n_pixels,
                    const uint32_t* minmax, int32_t mode,
                    int32_t round_dtype /*RGBA_F32 images: F16 rounds the value through binary16 first (a .half() module); else F32*/,
                    void* stream);

/* Shader.forward on tensors (rendering/shader.py:190-263): first pass for every mode.
 * Inputs (fp32) a mode does not read may be NULL.  frames: [N,3,3]; per-pixel arrays hold
 * n_pixels = N * pixels_per_camera entries.  image: [n_pixels,3] of image_dtype -- except for modes 1,2,5,
 * where it receives the fp32 first-pass values (image_dtype must be F32) and minmax + rm_shade_finish follow. */
int rm_shade_forward(const float* px_coords, const float* orientation, const float* frames, const float* dirs,
                     const float* coords, const float* normals, const float* lap, const float* dist,
                     void* image, int32_t image_dtype,
                     uint32_t* minmax, const void* cmap, int32_t cmap_size, int32_t cmap_dtype, int32_t mode,
                     int32_t degree, int64_t n_pixels, int64_t pixels_per_camera, void* stream);

/* VJP of rm_shade_forward for the per-pixel shaders that have one: Lambertian (0), vignette (3),
 * normal (4).  grad_image: [n_pixels, 1] (modes 0, 3) or [n_pixels, 3] (mode 4).  Outputs nullable. */
int rm_shade_backward(const float* dirs, const float* normals, const float* frames, const float* grad_image,
                      float* grad_dirs, float* grad_normals, int32_t mode, int64_t n_pixels,
                      int64_t pixels_per_camera, void* stream);

/* VJP of rm_render_forward w.r.t. scene parameters and, through the per-ray outputs, the camera pose, for all eight
 * shader modes.  0 lambertian, 3 vignette, 4 normal, 6 tangent, 7 spin: grad_image is dL/d(image) (6, 7: the colormap
 * index is piecewise constant; the brightness is differentiated).  1 distance, 2 proximity, 5 laplacian: the shader's
 * normalisation by the frame's minimum / maximum (shader.py:33-38, 51-55, 81-89) is a reduction over every pixel -- and
 * over every rank of a row-tiled render -- and is differentiated by the caller (ray_marching_amd/ops.py:
 * minmax_normalisation_vjp, laplacian_normalisation_vjp); grad_image[..., 0] must then hold dL/d(un-normalised value)
 * of the ray: of log(clamp(|origin - p|)), log(clamp(scene(p))) and the surface Laplacian.
 * fp32 only (cam->dtype must be RM_DTYPE_F32).  grad_image: device [N,rows,W,3].  grad_params[n_params] is overwritten.
 * work: nullable uint32[RM_WORK_WORDS] prepared by rm_minmax_init (dynamic tile queues);
 * flags: the RM_FLAG_TILE8X8 choice of the forward call; RM_FLAG_DYNAMIC_TILES; RM_FLAG_EARLY_OUT
 * stops a wave's reverse sweep once every ray's adjoint component along the ray is below the
 * rounding-error bound of its own dot product.
 * grad_pos / grad_dirs: when given, the per-ray gradients w.r.t. the world-frame ray origin and
 * direction are written (input of rm_camera_backward). */
int rm_render_backward(const RmScene* scene, const RmCamera* cam, const RmTetra* tetra,
                       const float* orientation, const float* translation,
                       const float* traj, const int32_t* nexec, const float* p_final,
                       const float* normal_u /*nullable: as written by the forward call (else recomputed)*/,
                       const float* grad_image, float* grad_params, float* partials, uint32_t* work,
                       float* grad_pos /*nullable [R,3]*/, float* grad_dirs /*nullable [R,3]*/,
                       float* grad_qdir /*nullable [R,4]: per-ray dL/d(orientation) through the shader's own use of the
                                          pose (modes 3, 6, 7); sum the rows of a camera with rm_sum_rows*/,
                       const void* cmap, int32_t cmap_size, int32_t cmap_dtype /*modes 6, 7: as in the forward call*/,
                       int32_t mode, int32_t degree, int32_t steps, int32_t row_begin, int32_t row_end, int32_t flags,
                       int32_t* tile_cost /*nullable out [T]: reverse march steps each tile's wave walked*/,
                       float* hard_ws /*nullable: rm_bwd_hard_floats(hard_capacity, steps) floats*/,
                       int64_t hard_capacity, void* stream);

/* VJP of the normalisation of the globally normalised shaders (modes 1 distance, 2 proximity: ((x - min) / (max - min))^(1/2.33),
 * shader.py:33-38, 51-55; mode 5 laplacian: clamp((x / max|x| * -1 + 1) / 2, 0, 1)^(1/2.33), shader.py:81-89), in the order
 * autograd walks it -- including the infinities and NaNs the reference's gradient carries at the rays of the frame's
 * minimum / maximum.  raw: the un-normalised values rm_render_forward left in `first_pass` ([n,3], channel 0 is read);
 * lohi: device {min, max} (rm_minmax_decode); grad_image: dL/d(image) [n,3]; grad_raw [n,3] out: channel 0 =
 * dL/d(un-normalised value), channels 1, 2 = 0 -- the `grad_image` rm_render_backward expects for these modes.
 * partials: RM_NORM_BWD_BLOCKS * 4 floats, 16-byte aligned.  Two launches (block sums, then the elementwise combination); deterministic. */
#define RM_NORM_BWD_BLOCKS 1024
int rm_shade_norm_backward(const float* raw, const float* grad_image, const float* lohi /*device [2]*/, int32_t mode,
                           float* grad_raw, float* partials, int64_t n_pixels, void* stream);

/* out[width] = sum over n_rows rows of rows[n_rows][width], fixed summation tree (deterministic). */
int rm_sum_rows(const float* rows, int64_t n_rows, int32_t width, float* out, void* stream);

/* Workspace of the deferred-ray path of rm_render_backward.  Rays whose march has not converged need a VJP at
 * every remaining step; walked by their own wave they are the critical path of the launch, so (with
 * RM_FLAG_EARLY_OUT and a `work` buffer) up to hard_capacity of them are put on a list and all their
 * (ray, step) pairs are evaluated in parallel by three follow-up kernels.  Rays beyond the capacity are walked
 * in place; hard_ws = NULL switches the path off.  Same gradients either way (to summation order).  hard_ws must be
 * 16-byte aligned. */
int64_t rm_bwd_hard_floats(int64_t capacity, int32_t steps);

/* VJP of PinholeCamera.forward (rendering/ray_marching.py:57-64) w.r.t. the pose: reduces per-ray
 * gradients grad_pos / grad_dirs ([N,rows,W,3], either may be NULL) to grad_orientation [N,4] and
 * grad_translation [N,3] (either may be NULL).  partials: N * RM_CAMERA_BWD_BLOCKS * 7 floats. */
#define RM_CAMERA_BWD_BLOCKS 256
int rm_camera_backward(const RmCamera* cam, const float* orientation, const float* grad_pos, const float* grad_dirs,
                       float* grad_orientation, float* grad_translation, float* partials, int32_t row_begin,
                       int32_t row_end, void* stream);

/* Host-side check of a compiled program BEFORE it is uploaded (host pointer):
 * opcode range, parameter/slot/derived offsets inside their blocks, balanced
 * begin/end and push/pop, stack depth <= stack_floats, exactly one result. */
int rm_validate_program(const int32_t* host_program, int32_t n_instr, int32_t n_params, int32_t n_derived,
                        int32_t stack_floats, int32_t n_slots);

#ifdef __cplusplus
}
#endif
#endif /* RM_ABI_H */

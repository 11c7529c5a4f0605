/*
 * relu_field.h -- C ABI of the MI355X (gfx950) ReLU-Fields render hot path.
 *
 * Shared library: thr3ed_atom_amd/csrc/librelu_field_hip.so (built by __graft_entry__.build()).
 *
 * The reference (akanimax/thr3ed_atom) is pure Python on PyTorch and has no FFI; its plug-in point for
 * this path is the Python callable type
 *     RenderProcedure = Callable[[Module, Rays, RenderConfig, Optional[int]], RenderOut]
 *     (thre3d_atom/thre3d_reprs/renderers.py:22-25), invoked only at
 *     thre3d_atom/modules/volumetric_model.py:112-114.
 * The entry points below are what a binding for that path binds to (INTEGRATION.md shows the ctypes stub);
 * each one names the reference code it replaces.
 *
 * Conventions
 *  - All pointers named *_dev are DEVICE pointers owned by the caller (PyTorch allocations); the library
 *    never allocates, frees or retains them.  All tensors are contiguous float32 unless a stride is given.
 *  - Every call only ENQUEUES work on `stream` (a hipStream_t passed as void*); no implicit synchronisation.
 *  - Re-entrant: no mutable state is shared between calls.  What the library keeps per process is read-only after its first
 *    use: tuning knobs read from the environment ($RF_FRAME_TILES, $RF_FWD_PAIR, $RF_EMIT_PAIR, $RF_BRICK_STAGGER, $RF_FAR_ADDRESSING -- A/B switches,
 *    unset in production) and the cached result of one-time hipFuncSetAttribute calls (dynamic LDS size of the brick kernels).
 *    Return value: RF_OK (0) or a negative RF_ERR_* code; nothing is thrown across the ABI.  rf_error_string() maps a code to text.
 *  - Gradient buffers are ACCUMULATED into (+=) with float32 hardware atomics; the caller zero-fills them
 *    (or keeps accumulating across several renders, which is autograd's semantics).
 */
#ifndef RELU_FIELD_H_
#define RELU_FIELD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RF_ABI_VERSION 4
/* brick_size value for bricks of 4 x 8 x 8 nodes (see the binned backward below) */
#define RF_BRICK_4X8X8 488

enum {
  RF_OK = 0,
  RF_ERR_NULL_POINTER = -1,
  RF_ERR_BAD_SHAPE = -2,
  RF_ERR_UNSUPPORTED = -3,
  RF_ERR_LAUNCH = -4
};

/* density activation variants of VoxelGrid (thre3d_reprs/voxels.py:292-309; the three configurations the
 * reference's trainer script builds: train_sh_based_voxel_grid_with_posed_images.py:169-192) */
enum {
  RF_DENSITY_RELU = 0,     /* pre = identity, post = ReLU      (the ReLU field)       */
  RF_DENSITY_SOFTPLUS = 1, /* pre = identity, post = softplus(beta=1, threshold=20)   */
  RF_DENSITY_ABS = 2,      /* pre = |.|,      post = identity  (traditional grid)     */
  RF_DENSITY_IDENTITY = 3  /* pre = identity, post = identity                         */
};

/* storage layout of the grid tensors handed to the kernels */
enum {
  RF_LAYOUT_REFERENCE = 0, /* the reference's two tensors: densities [X,Y,Z,1], features [X,Y,Z,F]
                              (feature index = colour*K + k, thre3d_reprs/voxels.py:70-71)                 */
  RF_LAYOUT_SPLIT = 1,     /* MI355X-native: densities_dev -> base [X,Y,Z,4] = (density, sh0 r, sh0 g, sh0 b),
                              features_dev -> rest [X,Y,Z,F-3] with index = colour*(K-1) + (k-1), k >= 1
                              (NULL when F == 3).  The diffuse pass and the density gather touch only the
                              16-byte base records; gradient buffers use the same layout.                 */
  RF_LAYOUT_BRICKED = 2    /* the split channel arrangement with BRICK-MAJOR node order: the grid is cut into
                              8x8x8-node bricks stored contiguously, base [NBX,NBY,NBZ,8,8,8,4] and rest
                              [NBX,NBY,NBZ,8,8,8,F-3] with NB* = ceil(dim/8) (nodes beyond dims are padding the
                              kernels never touch).  The 8 corners of a cell lie within ~1 KB (base) instead of
                              megabytes apart, and a brick of the binned backward is one contiguous write.    */
};

/* render flags (SHVoxGridRenderConfig fields, thre3d_reprs/renderers.py:28-45) */
enum {
  RF_FLAG_WHITE_BKGD = 1,     /* white_bkgd                                                        */
  RF_FLAG_RENDER_DIFFUSE = 2, /* render_diffuse: only the degree-0 SH coefficient of each colour   */
  RF_FLAG_AABB_SAMPLING = 4,  /* optimized_sampling: per-ray [t_enter, t_exit] from the slab test  */
  RF_FLAG_OCCUPANCY_SKIP = 8, /* use RFGrid.occupancy_dev to skip provably-empty cells (exact)     */
  RF_FLAG_JITTER_KEYED = 16   /* perturb_sampled_points without a [N,S] tensor: when RFRayBatch.t_rand_dev is NULL the jitter
                                 of (ray, sample) is a counter-based hash of (jitter_key, first_ray + ray, sample) -- the law of
                                 torch.rand (uniform on [0,1), independent), evaluated where it is used                    */
};

/* The dense SH + density voxel grid: VoxelGrid (thre3d_reprs/voxels.py:46-331). */
typedef struct RFGrid {
  const float* densities_dev; /* [X, Y, Z, 1]; z fastest                                              */
  const float* features_dev;  /* [X, Y, Z, F]; F = 3 * (deg+1)^2, channel-major: index = colour*K + k  */
  int32_t dims[3];            /* X, Y, Z                                                               */
  int32_t num_features;       /* F in {3, 12, 27, 48}                                                  */
  int64_t density_stride;     /* floats between consecutive voxels of densities_dev (1 = reference)   */
  int64_t feature_stride;     /* floats between consecutive voxels of features_dev  (F = reference)   */
  float aabb_min[3];          /* float32 planes of the bounding box (voxels.py:187-212)                */
  float aabb_max[3];
  float norm_scale[3];        /* q = p * scale + bias maps the AABB to [-1, 1] (voxels.py:214-223,     */
  float norm_bias[3];         /*   utils/imaging_utils.py:58-63); computed by the host in float32      */
  float density_scale;        /* expected_density_scale rho, applied BEFORE interpolation              */
  int32_t density_mode;       /* RF_DENSITY_*                                                          */
  int32_t layout;             /* RF_LAYOUT_*; strides above are then those of base / rest              */
  const uint32_t* occupancy_dev; /* optional bit mask, one bit per cell incl. the border cells:
                                    (X+1)*(Y+1)*(Z+1) bits, see rf_build_occupancy; may be NULL      */
} RFGrid;

/* A posed pinhole camera (CameraIntrinsics + CameraPose, utils/imaging_utils.py:17-30) for rays generated in-kernel:
 * cast_rays (rendering/volumetric/utils/misc.py:12-50) fused into the render.  HOST struct. */
typedef struct RFCamera {
  int32_t height, width;
  float focal;
  float pose[12]; /* [3,4] row-major = rotation | translation, camera-to-world */
} RFCamera;

/* Flat rays + sampling parameters: Rays (rendering/volumetric/render_interface.py:13-44),
 * sample_uniform_points_on_rays (rendering/volumetric/sample.py:15-68). */
typedef struct RFRayBatch {
  const float* origins_dev;    /* [N, 3]   (may be NULL when `camera` is given)                        */
  const float* directions_dev; /* [N, 3], not normalised                                               */
  int64_t num_rays;            /* N                                                                    */
  int32_t num_samples;         /* S >= 1                                                               */
  float near;                  /* CameraBounds.near / far (float32)                                    */
  float far;
  const float* t_vals_dev;     /* [S] = linspace(0, 1, S) (sample.py:46)                               */
  const float* t_rand_dev;     /* [N, S] jitter in [0,1) (sample.py:63) or NULL = perturb off / keyed  */
  uint64_t jitter_key;         /* RF_FLAG_JITTER_KEYED: key of the counter-based jitter                */
  int64_t first_ray;           /* global index of ray 0 of this batch: position in the keyed jitter stream, and the
                                  pixel (row-major) of ray 0 when `camera` generates the rays -- chunks of one frame
                                  rendered with first_ray = chunk offset give the frame's rays bit for bit      */
  const RFCamera* camera;      /* optional HOST pointer: rays r = pixel-centre rays of pixels first_ray + r of this
                                  camera (utils/misc.py:12-50); origins_dev / directions_dev are then ignored   */
} RFRayBatch;

/* Per-ray outputs: RenderOut (render_interface.py:47-83) + extra {"disparity", "accumulated_weight"}. */
typedef struct RFRenderOut {
  float* colour_dev;    /* [N, 3]                                                                      */
  float* depth_dev;     /* [N]                                                                         */
  float* acc_dev;       /* [N]                                                                         */
  float* disparity_dev; /* [N]   (NaN where acc == 0, like the reference)                              */
  /* Optional per-sample cache written by the forward pass and consumed by rf_render_backward* (all four NULL for
   * inference).  Only the samples that can carry gradient are cached -- inside the box, T != 0, and sigma != 0 under ReLU
   * (every other sample contributes exactly nothing to any adjoint) -- COMPACTED per chunk of 64 samples: the j-th such
   * sample of chunk c of ray r is entry r * S + 64 c + j, and bit l of chunk_mask_dev[r][c] says that sample 64 c + l is
   * one of them.  Everything else in the two caches is left unwritten (and unread). */
  float* sample_cache_dev; /* [N, S, 4] = (raw r, raw g, raw b, sigma)                                 */
  float* trans_cache_dev;  /* [N, S]    = transmittance T_i                                            */
  int32_t* stop_cache_dev; /* [N]       = number of samples the forward pass processed                 */
  uint64_t* chunk_mask_dev; /* [N, ceil(S / 64)] which samples of each chunk are cached               */
  /* Optional (with the cache): the forward pass also COUNTS the cached samples per (brick, flags) key of the binned
   * backward (see rf_render_backward_emit) into key_hist_dev [8 * nbricks] (added to; clear before the first use), so
   * that rf_render_backward_emit_direct can write their records straight to the final positions. */
  int32_t* key_hist_dev;
  int32_t brick_size;      /* 4, 8 or RF_BRICK_4X8X8 (only read when key_hist_dev != NULL)             */
} RFRenderOut;

/* Upstream gradients of a render (all [N, ...] device arrays; any may be NULL = zero). */
typedef struct RFRenderGrads {
  const float* grad_colour_dev; /* [N, 3] */
  const float* grad_depth_dev;  /* [N]    */
  const float* grad_acc_dev;    /* [N]    */
} RFRenderGrads;

int rf_abi_version(void);
const char* rf_error_string(int code);
/* sizeof() of the ABI structs as this library was compiled, so that a binding can verify its own mirrors before the first
 * call: which = 0 RFGrid, 1 RFRayBatch, 2 RFRenderOut, 3 RFRenderGrads, 4 RFBrickList, 5 RFAdamState, 6 RFCamera,
 * 7 RFRaySelection, 8 RFPassScratch, 9 RFTrainStep; -1 for any other value. */
int rf_abi_struct_size(int which);

/* cast_rays (rendering/volumetric/utils/misc.py:12-50) + flatten_rays (:53-57):
 * all H*W pixel-centre rays of one camera, row-major (ray = i*W + j).  rotation/translation are HOST
 * pointers to 9 / 3 floats (camera-to-world). */
int rf_cast_rays(int32_t height, int32_t width, float focal, const float* rotation_host,
                 const float* translation_host, float* origins_dev, float* directions_dev, void* stream);

/* Which kernel rf_render_forward uses for a frame of this camera (RFRayBatch.camera set, no sample cache; the frame loop of
 * modules/volumetric_model.py:143-172): 1 = ray packets (one wavefront per 8 x 8 pixel tile, the tile's 4 x 4 x 4-node neighbourhood
 * staged through LDS once per sample index), 0 = one wavefront per ray; negative = error code.  The packet kernel is chosen where a
 * tile's footprint at the volume's centre stays within 2 voxels (3 with RF_FLAG_OCCUPANCY_SKIP and a mask), on split / bricked
 * storage with a 16-byte aligned base tensor, every SH degree, keyed jitter or none (a frame with a jitter TABLE goes to the per-ray
 * kernel) and at most 1024 samples per ray (this function does not see the sample count: a frame of more goes to the per-ray kernel
 * whatever it answers -- a packet lane adds a ray's samples sequentially, and that float32 sum drifts past the parity bar at 4096+);
 * $RF_FRAME_TILES = 1 / 0 in the environment forces / forbids it within these rules.  Both kernels compute a pixel with the
 * same per-sample arithmetic and differ in the order a ray's weighted samples are added (<= 2e-6 on colours).  Host-side only: no
 * device access.  (Added to ABI version 4 compatibly: no existing struct or signature changed.) */
int rf_frame_render_kernel(const RFGrid* grid, const RFCamera* camera, uint32_t flags);

/* The training-iteration ray source (modules/trainers.py:281-303): rays of the selected pixels only.
 * pixel_index_dev[r] indexes the concatenation of B images: b*H*W + i*W + j; poses_dev is [B, 3, 4]
 * (rotation | translation) on the device. */
int rf_cast_selected_rays(int32_t height, int32_t width, float focal, const float* poses_dev,
                          int32_t num_poses, const int64_t* pixel_index_dev, int64_t num_rays,
                          float* origins_dev, float* directions_dev, void* stream);

/* sample_random_rays_and_pixels_synchronously (rendering/volumetric/utils/misc.py:117-129) fused with the ray
 * casting and pixel gathering around it (modules/trainers.py:281-303): element r of the batch is pixel
 * PRP_key(r) of the num_batch_images*H*W pixels of the image batch, PRP a keyed bijection of that range
 * (distinct pixels, uniformly distributed: the law of torch.randperm(P)[:R], without sorting P keys).
 * poses_dev [M,3,4] and pixel_table_dev [M*H*W,3] describe ALL images; image_ids_dev [num_batch_images]
 * (int64, may be NULL = images 0..B-1) picks the batch.  pixel_index_dev [R] (may be NULL) receives
 * b*H*W + i*W + j. */
int rf_select_rays_and_pixels(int32_t height, int32_t width, float focal, const float* poses_dev,
                              const int64_t* image_ids_dev, int32_t num_batch_images, const float* pixel_table_dev,
                              uint64_t key, int64_t first_index, int64_t num_rays, float* origins_dev,
                              float* directions_dev, float* pixels_dev, int64_t* pixel_index_dev, void* stream);

/* _ray_aabb_intersection (rendering/volumetric/sample.py:71-184): bounds_dev [N, 2], hit_dev [N] (0/1,
 * may be NULL). */
int rf_ray_aabb_bounds(const float* origins_dev, const float* directions_dev, int64_t num_rays, float near,
                       float far, const float* aabb_min_host, const float* aabb_max_host,
                       float* bounds_dev, float* hit_dev, void* stream);

/* render_sh_voxel_grid (thre3d_reprs/renderers.py:48-102) = sampler (sample.py) -> VoxelGrid.forward
 * (voxels.py:276-331) -> SH (utils/spherical_harmonics.py:64-116) -> AABB mask (process.py:80-84) ->
 * compositing (accumulate.py:31-113), fused in one launch. */
int rf_render_forward(const RFGrid* grid, const RFRayBatch* rays, uint32_t flags, const RFRenderOut* out,
                      void* stream);

/* The adjoint of rf_render_forward into the grid (the reference gets it from autograd:
 * grid_sampler_3d_backward, cumprod_backward, ...).  `fwd` must hold the caches written by the matching
 * forward call.  grad_densities_dev [X,Y,Z,1] and grad_features_dev [X,Y,Z,F] are accumulated into. */
int rf_render_backward(const RFGrid* grid, const RFRayBatch* rays, uint32_t flags, const RFRenderOut* fwd,
                       const RFRenderGrads* grads, float* grad_densities_dev, float* grad_features_dev,
                       void* stream);

/* ---- binned backward: the same adjoint as rf_render_backward without float atomics ----------------------------------
 * The atomic scatter of rf_render_backward is bound by the memory-side atomic unit.  The binned variant turns every
 * contributing sample into a RECORD, puts the records in (brick, flags) order and lets one workgroup per brick sum them
 * on chip under exclusive ownership:
 *
 *   key  = ((((bx * 2 + f_x) * NBY + by) * NBZ + bz) << 2) | f_y | f_z << 1;  (bx, by, bz) = the brick (brick_size^3 nodes,
 *          brick_size in {4, 8}; or RF_BRICK_4X8X8: 4 x 8 x 8 nodes, the bricks of the single-GPU optimizer pass -- four
 *          256-thread workgroups per CU instead of two 512-thread ones; accepted by the render / emit / offset functions and
 *          by rf_brick_accumulate_adam and rf_train_step, SH degree 0 or 2, every grid tensor below 2^30 elements)
 *          holding the LOWER node of the sample's cell;  f_a = the cell's upper node on axis a belongs
 *          to the next brick (so the record also touches that neighbour's nodes).  8 * nbricks keys.  The order is x-slab
 *          major with the x flag directly below the slab index: everything that touches the nodes of the x-slabs [s0, s1)
 *          of bricks is ONE contiguous key range, [key(s0 - 1, f_x = 1), key(s1, f_x = 0)) -- what a data-parallel rank
 *          sends the owner of those slabs is a single slice of its sorted list (see rf_brick_accumulate_adam_range).
 *   record = rf_expanded_record_floats(F) floats, COMPACT:
 *          F > 3:  (continuous index x, y, z, dL/d density) (dL/d raw r, g, b, v_x) (v_y, v_z, 0, 0) -- 48 B; v = the ray's
 *                  unit viewing direction.  The per-channel values dL/d raw[colour] * Y_k(v) of a node's 3K + 1 channels are
 *                  expanded by rf_brick_accumulate in LDS (the reference's evaluate_spherical_harmonics operation order):
 *                  they never exist in HBM.
 *          F == 3, and every list of a render_diffuse pass whatever the grid's degree:  (index x, y, z, 0) (dL/d density,
 *                  dL/d sh0 r, g, b) -- 32 B.
 *   offsets [8 * nbricks + 1] (int64): start of every key class in the record list (last = end).
 *
 * Three front ends produce (records in key order, offsets); rf_brick_accumulate consumes them:
 *   A. fused (default of the trainer): rf_render_forward with RFRenderOut.key_hist_dev COUNTS the records per key ->
 *      rf_bin_offsets -> rf_render_backward_emit_direct writes each record at the next free position of its key.
 *      No per-slot arrays; integer atomics only (counters, cursors); any number of bricks up to 2^18.
 *   B. counting sort after the fact: rf_render_backward_emit (per-slot 16-bit keys + per-slot records, hist_dev) ->
 *      rf_bin_offsets -> rf_scatter_records.
 *   C. deterministic: rf_render_backward_emit -> torch.sort of the keys (stable radix) + searchsorted ->
 *      rf_expand_records.  Fixed float32 summation order: bit-reproducible gradients.
 *   B and C use 16-bit keys: at most 4096 bricks.  The order inside a key class depends on atomic timing in A and B.
 *
 * rf_brick_accumulate: one workgroup per brick OWNS the brick's nodes: it reads exactly the key classes that touch them
 * (its own and, per flags, up to 7 lower neighbours'), sums them in MFMA accumulators (node sums = trilinear weights x the
 * records' channel values, four records per v_mfma_f32_16x16x4_f32: exact float32 fma chains, no atomics of any kind,
 * summation order fixed by the record order) and writes the brick with plain coalesced stores: accumulate = 0 OVERWRITES every
 * element of the gradient tensors (no zero-fill needed; diffuse lists: only density + degree-0 gradients), accumulate = 1
 * adds.  Up to 16 lists per call, at most 8 of a kind, the full-width lists first: (specular list, render_diffuse list) = BOTH
 * renders of a training iteration (modules/trainers.py:306-341) in one pass (the base-channel records go into the first four
 * channel columns of the same accumulators); under data parallelism one pair per source rank.  Any SH degree (degree 3: brick_size 8). */
typedef struct RFBrickList {
  const float* records_sorted_dev; /* [capacity, rf_expanded_record_floats(F)] (diffuse lists: F = 3) */
  const int64_t* offsets_dev;      /* [8 * nbricks + 1] start of each (brick, flags) class            */
  int32_t render_diffuse;          /* records come from a render_diffuse pass                         */
} RFBrickList;

int32_t rf_expanded_record_floats(int32_t num_features);

/* exclusive prefix sums of the per-key counters hist_dev [num_keys] -> offsets_dev [num_keys + 1] (positions start at 0)
 * and an int32 copy cursor_dev [num_keys] for the atomic cursors of A / B */
int rf_bin_offsets(const int32_t* hist_dev, int32_t num_keys, int64_t* offsets_dev, int32_t* cursor_dev, void* stream);

/* A: every sample counted (= cached) by the forward call that produced `fwd` writes its record -- zeros if its gradient
 * happens to vanish -- at the next free position of its key; hist_clear_dev
 * [8 * nbricks] (may be NULL) is cleared for the next iteration. */
int rf_render_backward_emit_direct(const RFGrid* grid, const RFRayBatch* rays, uint32_t flags, const RFRenderOut* fwd,
                                   const RFRenderGrads* grads, int32_t brick_size, int32_t* cursor_dev,
                                   float* records_sorted_dev, int32_t* hist_clear_dev, void* stream);

/* B / C: per contributing sample its record (format above) at its SLOT: records_dev [N*S, rf_expanded_record_floats(F)]
 * (render_diffuse passes: F = 3), and, for EVERY slot, keys_dev [N*S] (-1 = no gradient).  hist_dev [8 * nbricks] (may be NULL;
 * zero before the first use) is incremented by the number of records per key (B). */
int rf_render_backward_emit(const RFGrid* grid, const RFRayBatch* rays, uint32_t flags, const RFRenderOut* fwd,
                            const RFRenderGrads* grads, int32_t brick_size, int16_t* keys_dev, float* records_dev,
                            int32_t* hist_dev, void* stream);
/* B: every keyed slot's record goes to the next free position of its key; clears hist_dev (may be NULL) */
int rf_scatter_records(const RFGrid* grid, const int16_t* keys_dev, const float* records_dev, int64_t capacity,
                       int32_t* cursor_dev, int32_t render_diffuse, float* records_sorted_dev, int32_t* hist_dev, int32_t num_keys,
                       void* stream);
/* C: records_sorted[i] = record of slot perm_dev[i] for *begin_dev (= offsets[0]: unkeyed slots sort in front) <= i < capacity */
int rf_expand_records(const RFGrid* grid, const float* records_dev, const int64_t* perm_dev, const int64_t* begin_dev,
                      int64_t capacity, int32_t render_diffuse, float* records_sorted_dev, void* stream);

int rf_brick_accumulate(const RFGrid* grid, int32_t brick_size, const RFBrickList* lists, int32_t num_lists,
                        float* grad_densities_dev, float* grad_features_dev, int32_t accumulate, void* stream);

/* The optimizer of the training iteration (torch.optim.Adam(betas, eps), modules/trainers.py:242-250,339-341) fused into the
 * brick pass: the workgroup that owns a brick holds the COMPLETE gradient of its parameters on chip (when the lists carry
 * every render of the iteration), so it applies the Adam update right there -- the gradient tensor never exists in HBM and
 * the separate optimizer pass (7 x 4 B per parameter) becomes 6 x 4 B inside this one.  Same arithmetic as rf_adam_step
 * (hardware square root and reciprocal, 1 ulp).
 * Requirements: RF_LAYOUT_SPLIT / RF_LAYOUT_BRICKED with F in {3, 27} (whole float4s per node), all six pointers 16-byte
 * aligned, param_*_dev the very tensors `grid` describes.  Bricks without records still take their (zero-gradient) step. */
typedef struct RFAdamState {
  float* param_first_dev;      /* = grid->densities_dev (base)  */
  float* param_second_dev;     /* = grid->features_dev  (rest; NULL when F == 3) */
  float* exp_avg_first_dev;    /* first moments, same shapes    */
  float* exp_avg_second_dev;
  float* exp_avg_sq_first_dev; /* second moments                */
  float* exp_avg_sq_second_dev;
  float lr, beta1, beta2, eps;
  int32_t step;                /* 1-based step count (bias correction) */
} RFAdamState;

int rf_brick_accumulate_adam(const RFGrid* grid, int32_t brick_size, const RFBrickList* lists, int32_t num_lists,
                             const RFAdamState* adam, void* stream);

/* The same pass (whole grid, brick_size == RF_BRICK_4X8X8) for a grid held in the REFERENCE's two tensors whose forward passes run
 * from a split-layout shadow: `grid` / `adam` describe the shadow (RF_LAYOUT_SPLIT, F in {3, 27}), and the flush writes every
 * updated parameter a second time, in the reference's own layout -- mirror_densities_dev [X,Y,Z,1], mirror_features_dev [X,Y,Z,F]
 * with feature index = colour * K + k (thre3d_atom/thre3d_reprs/voxels.py:70-71, rendering/volumetric/process.py:61,66), both
 * contiguous and 16-byte aligned -- so that the nn.Parameters a reference user holds stay in sync with the shadow without a
 * re-layout launch (rf_convert_grid: 235 MB read + 235 MB written per iteration at 128^3 / SH degree 2; here +235 MB written out
 * of LDS).  Grid dims must be multiples of (4, 8, 8); RF_ERR_UNSUPPORTED otherwise (use rf_brick_accumulate_adam, then
 * rf_convert_grid).  This is what optimizer.step() of modules/trainers.py:341 becomes for the strict drop-in.  (Added to ABI
 * version 4 compatibly: no existing struct or signature changed.) */
int rf_brick_accumulate_adam_mirror(const RFGrid* grid, int32_t brick_size, const RFBrickList* lists, int32_t num_lists,
                                    const RFAdamState* adam, float* mirror_densities_dev, float* mirror_features_dev, void* stream);

/* The same pass restricted to the bricks [first_brick, first_brick + num_bricks) (brick id = (bx * NBY + by) * NBZ + bz): the
 * OWNER-COMPUTES step of data-parallel training.  The reference trains on one device (modules/trainers.py:338-341 is its
 * loss.backward(); optimizer.step()); with N ranks each rank owns a range of x-slabs of bricks, receives from every rank the
 * slice of that rank's sorted record lists that touches its slabs (one contiguous key range, see the key order above; the
 * lists here are then one (specular, render_diffuse) pair per source rank, each with that rank's offsets table and a records
 * pointer positioned such that records_sorted_dev + offsets[k] * record size is the first record of key k), sums them and
 * applies Adam to its own parameters only; an all-gather of the parameters follows.  Scale the losses by 1 / N
 * (RFTrainStep.loss_scale) so that the sum over the ranks' records is the mean gradient. */
int rf_brick_accumulate_adam_range(const RFGrid* grid, int32_t brick_size, const RFBrickList* lists, int32_t num_lists,
                                   const RFAdamState* adam, int32_t first_brick, int32_t num_bricks, void* stream);

/* The owner's pass with SEVERAL workgroups per brick.  With N ranks an owned brick receives N times the records of the single-GPU
 * case while the launch covers only 1 / N of the bricks: one workgroup per brick leaves most of the machine idle behind the
 * heaviest bricks (measured at N = 8 on the bench workload: 0.35 ms for the x-slab through the middle of the volume, half of the
 * 512 workgroup slots unused).  Here `parts` (1..8) workgroups share a brick: workgroup (brick, part) sums the lists l of each kind
 * with l % parts == part -- the source ranks are dealt out --, the partial accumulator images meet in `scratch_dev`, and the last
 * workgroup of a brick to arrive adds them and applies Adam (no workgroup waits for another).  scratch_dev: caller-owned,
 * rf_brick_split_scratch_bytes(grid, num_bricks, parts) bytes, 16-byte aligned, ZERO before the first launch; every launch leaves
 * its counters zero again (launches that share a scratch buffer must not overlap).  Needs 8^3 bricks and the one-round flush (the
 * grid tensors below 2^30 elements); RF_ERR_UNSUPPORTED otherwise (use rf_brick_accumulate_adam_range).  (Added to ABI
 * version 4 compatibly: no existing struct or signature changed.) */
int64_t rf_brick_split_scratch_bytes(const RFGrid* grid, int32_t num_bricks, int32_t parts);
int rf_brick_accumulate_adam_split(const RFGrid* grid, int32_t brick_size, const RFBrickList* lists, int32_t num_lists,
                                   const RFAdamState* adam, int32_t first_brick, int32_t num_bricks, int32_t parts, void* scratch_dev,
                                   int64_t scratch_bytes, void* stream);

/* VoxelGrid.forward (thre3d_reprs/voxels.py:276-331) as a standalone point query: points_dev [M,3] (any
 * points: zeros padding outside the grid, no AABB mask) -> out_dev [M, F+1] = (F interpolated features in the
 * reference order colour*K + k, activated density).  Bit-for-bit the ATen grid_sample recipe. */
int rf_grid_query(const RFGrid* grid, const float* points_dev, int64_t num_points, float* out_dev, void* stream);

/* Its adjoint into the grid tensors (accumulates, storage layout of `grid`); grad_out_dev is [M, F+1]. */
int rf_grid_query_backward(const RFGrid* grid, const float* points_dev, int64_t num_points,
                           const float* grad_out_dev, float* grad_densities_dev, float* grad_features_dev,
                           void* stream);

/* Exact empty-cell mask for RF_FLAG_OCCUPANCY_SKIP (BASELINE.json configs[4]):
 * bit (cx, cy, cz), cx in [0, X] etc., is set iff any of the (up to 8) grid nodes
 * (cx-1..cx, cy-1..cy, cz-1..cz) that exist has a raw density that can yield sigma != 0 under
 * `density_mode` (ReLU: D*rho > threshold; other modes: every cell is occupied).  With threshold = 0 and
 * ReLU the skip is bit-exact.  occupancy_dev holds ceil((X+1)(Y+1)(Z+1)/32) words. */
int rf_build_occupancy(const RFGrid* grid, float threshold, uint32_t* occupancy_dev, void* stream);

/* Stage transition of the trainer: scale_voxel_grid_with_required_output_size (thre3d_reprs/voxels.py:334-373), i.e.
 * F.interpolate(mode="trilinear", align_corners=False) of the whole [F+1]-channel volume, from `src` into the tensors `dst`
 * describes (any dims, any layout on either side, same num_features; dst's tensors are overwritten; dst's AABB / activation
 * fields are not used).  ATen's index and weight arithmetic and the summation order of its channels-last CPU kernel (8-wide
 * vector body, scalar tail): bit-identical to what the reference's function returns on the CPU.  (Added to ABI version 3
 * compatibly: no existing struct or signature changed.) */
int rf_upsample_grid(const RFGrid* src, const RFGrid* dst, void* stream);

/* Re-layout: every (node, channel) of `src` copied into the tensors `dst` describes (same dims and num_features, any layouts;
 * dst's tensors are overwritten, its AABB / activation fields are not used).  A binding that must keep the reference's own two
 * Parameters (thre3d_reprs/voxels.py:70-71) can keep an RF_LAYOUT_SPLIT shadow of them for the forward passes this way. */
int rf_convert_grid(const RFGrid* src, const RFGrid* dst, void* stream);

/* The loss of the training iteration (modules/trainers.py:311-317, 329-336) in one launch:
 * grad_colour_dev [N,3] = scale * d(mean |colour - target|)/d colour = scale * sign(colour - target) / (3N);
 * sums_dev[0] += sum |colour - target|, sums_dev[1] += sum (colour - target)^2  (L1 loss and MSE/PSNR for
 * logging; the caller zeroes sums_dev). */
int rf_l1_loss_grad(const float* colour_dev, const float* target_dev, int64_t num_rays, float scale,
                    float* grad_colour_dev, float* sums_dev, void* stream);

/* Fused Adam step on a flat float32 parameter buffer (torch.optim.Adam semantics, betas/eps/lr given;
 * modules/trainers.py:242-250,339-341).  step is the 1-based step count used for bias correction.
 * zero_grad != 0 also clears grad_dev after reading it (optimizer.zero_grad() of the next iteration). */
int rf_adam_step(float* param_dev, float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t numel,
                 float lr, float beta1, float beta2, float eps, int32_t step, int32_t zero_grad, void* stream);

/* ---- the training iteration as ONE call -------------------------------------------------------------------------------
 * modules/trainers.py:278-341 (ray/pixel subset -> specular render -> L1 -> diffuse render -> L1 -> backward of both ->
 * Adam) enqueued by a single host call: select -> forward (counts records), twice -> loss gradients + offsets -> emit, twice
 * -> ONE brick pass over both record lists [-> Adam inside its flush].  Seven launches, no host round trip between them; the
 * caller (a Python trainer, or any host language) pays one FFI crossing per iteration instead of ~20.  Every buffer is
 * caller-owned scratch that can be reused from step to step. */
typedef struct RFRaySelection { /* arguments of rf_select_rays_and_pixels */
  int32_t height, width;
  float focal;
  const float* poses_dev;
  const int64_t* image_ids_dev;
  int32_t num_batch_images;
  const float* pixel_table_dev;
  uint64_t key;
  int64_t first_index;
} RFRaySelection;

typedef struct RFPassScratch { /* one per render of the iteration: [0] specular, [1] render_diffuse */
  RFRenderOut out;           /* all eight per-ray / per-sample buffers, key_hist_dev [8 * nbricks] (zero on entry; left
                                zero) and brick_size                                                              */
  float* grad_colour_dev;    /* [N, 3]                                                                           */
  int32_t* cursor_dev;       /* [8 * nbricks]                                                                    */
  int64_t* offsets_dev;      /* [8 * nbricks + 1]                                                                */
  float* records_sorted_dev; /* [N * S, rf_expanded_record_floats(F)] ([1]: rf_expanded_record_floats(3))        */
  const float* t_rand_dev;   /* [N, S] jitter of this render, or NULL (off, or keyed with RF_FLAG_JITTER_KEYED)  */
  uint64_t jitter_key;
} RFPassScratch;

typedef struct RFTrainStep {
  const RFRaySelection* select; /* non-NULL: the batch is drawn by the fused selection INTO origins / directions / pixels;
                                   NULL: they are inputs                                                          */
  float* origins_dev;           /* [N, 3] */
  float* directions_dev;        /* [N, 3] */
  float* pixels_dev;            /* [N, 3] target colours */
  int64_t num_rays;
  int32_t num_samples;
  float near, far;
  const float* t_vals_dev;      /* [S] */
  uint32_t flags;               /* RF_FLAG_* of both renders (RF_FLAG_RENDER_DIFFUSE is added to the second one here) */
  RFPassScratch pass[2];
  float* loss_sums_dev;         /* [4], overwritten: (sum |d|, sum d^2) of the specular render, then of the diffuse render */
  const RFAdamState* adam;      /* optimizer fused into the brick flush, or NULL: the gradient of L1 + L1 is WRITTEN (not
                                   added) to grad_first_dev / grad_second_dev (layout of `grid`)                  */
  float* grad_first_dev;
  float* grad_second_dev;
  int64_t first_ray;            /* position of ray 0 in the keyed jitter streams (global-batch data parallelism: the rank's
                                   offset in the batch); 0 otherwise                                                          */
  uint32_t phases;              /* 0 = the whole iteration, else a set of RF_STEP_FORWARD (select, both forward passes, losses +
                                   offsets), RF_STEP_EMIT (both adjoints as records), RF_STEP_BRICKS (the brick pass over
                                   pass[0..1]'s lists): a data-parallel caller starts its exchange of the offset tables after
                                   FORWARD, lets it overlap EMIT, and runs rf_brick_accumulate_adam_range itself after the
                                   record exchange                                                                              */
  float loss_scale;             /* the L1 gradients are scaled by this (0 = 1): 1 / world size under data parallelism          */
  void* const* timing_events;   /* optional HOST array of RF_TRAIN_STEP_EVENTS hipEvent_t (created by the caller with timing
                                   enabled): event 0 is recorded on `stream` before the first launch, event k after launch k
                                   in the order select, forward[0], (nothing), forward[1], losses of both renders + offsets of
                                   both lists (one launch), (nothing), emit[0], (nothing), emit[1], bricks -- per-kernel durations
                                   of the very call that is timed (only with phases == 0).  A full iteration runs both forward
                                   renders in ONE launch and both adjoints in ONE launch (the render_diffuse pass walks the same
                                   rays as the specular one: it finds the base records of its corners on chip); slot [0] of a
                                   pair then holds the launch and slot [1] is empty.  $RF_FWD_PAIR=0 / $RF_EMIT_PAIR=0 in the
                                   environment restore one launch per render (A/B measurements)                                */
} RFTrainStep;

enum {
  RF_STEP_FORWARD = 1, RF_STEP_EMIT = 2, RF_STEP_BRICKS = 4,
  RF_STEP_EMIT_SPECULAR = 8, RF_STEP_EMIT_DIFFUSE = 16,  /* one adjoint at a time */
  /* RF_STEP_FORWARD in two pieces, the render_diffuse pass first: it reads the base tensor only, so a data-parallel caller runs it
     while the `rest` parameters of the previous iteration are still being all-gathered */
  RF_STEP_SELECT_AND_DIFFUSE_FORWARD = 32, RF_STEP_SPECULAR_FORWARD_AND_LOSSES = 64,
  /* the two renders' chains apart (pipelined owner-computes step): RF_STEP_DIFFUSE_CHAIN = selection, render_diffuse forward, loss +
     offsets of list [1], its adjoint as records -- everything that reads the base tensor only and needs no other rank, so it runs
     beside the arriving `rest` parameters of the previous iteration; RF_STEP_SPECULAR_FORWARD = specular forward, loss + offsets of
     list [0] (then RF_STEP_EMIT_SPECULAR).  (Added compatibly: no struct or signature changed.) */
  RF_STEP_DIFFUSE_CHAIN = 128, RF_STEP_SPECULAR_FORWARD = 256
};

#define RF_TRAIN_STEP_EVENTS 11

int rf_train_step(const RFGrid* grid, const RFTrainStep* step, void* stream);

/* ---- the two renders of an iteration, PAIRED, as separate calls ---------------------------------------------------------
 * The reference's iteration renders the same rays twice -- vol_mod.render_rays(rays) and vol_mod.render_rays(rays,
 * render_diffuse=True), each with its own jitter draw (modules/trainers.py:306, 323-325) -- and back-propagates the sum of the two
 * L1 losses (:311, 329-330, 338).  A host that keeps autograd in charge (torch.autograd.Function: one node for the PAIR of
 * renders, one for the pair of losses) enqueues the iteration's launches one by one instead of through rf_train_step:
 *
 *   rf_render_forward_pair              both saving forward renders in ONE launch ([0] specular, [1] render_diffuse; the second
 *                                       finds the base records of its corners on chip).  Both outs must carry the four cache
 *                                       buffers; key_hist_dev counts the records per key as in rf_render_forward.
 *   rf_l1_loss_grad_pair                the two losses in ONE launch and ready to use: out_dev[5] = (loss[0] + loss[1], loss[0],
 *                                       mse[0], loss[1], mse[1]) -- means, written by the last workgroup to finish;
 *                                       grad_colour_dev[i] as rf_l1_loss_grad.  workspace_dev: 8 floats, zero before the first
 *                                       use, left zero (the kernel cleans up after itself; one workspace per stream).
 *   rf_bin_offsets_pair                 rf_bin_offsets for both lists in one launch
 *   rf_render_backward_emit_direct_pair both adjoints as records in ONE launch (rf_render_backward_emit_direct twice); upstream
 *                                       gradients of the colours only (passes[i].grad_colour_dev), like every reference use.
 *
 * Every array argument holds TWO entries.  RF_ERR_UNSUPPORTED when the two renders do not pair up (different ray counts, flags
 * that do not say specular / render_diffuse): the caller then launches them one by one.  (Added to ABI version 4 compatibly: no
 * existing struct or signature changed.) */
int rf_render_forward_pair(const RFGrid* grid, const RFRayBatch* rays, const uint32_t* flags, const RFRenderOut* outs, void* stream);
int rf_l1_loss_grad_pair(const float* const* colour_dev, const float* target_dev, int64_t num_rays, float scale,
                         float* const* grad_colour_dev, float* workspace_dev, float* out_dev, void* stream);
int rf_bin_offsets_pair(const int32_t* const* hist_dev, int32_t num_keys, int64_t* const* offsets_dev, int32_t* const* cursor_dev,
                        void* stream);
int rf_render_backward_emit_direct_pair(const RFGrid* grid, const RFRayBatch* rays, const uint32_t* flags,
                                        const RFPassScratch* passes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RELU_FIELD_H_ */

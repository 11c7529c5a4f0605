// relu_field_kernels.hip -- hand-written CDNA4 (gfx950, wave64) kernels of the ReLU-Fields render path.
//
// One wavefront renders one ray.  A ray is processed in chunks of 64 samples; each chunk runs three
// phases that re-shape the wave instead of launching separate kernels:
//
//   P0  lanes = samples   z, point, AABB test, 8-corner density gather (+rho, +ReLU), alpha, wave-level
//                         exclusive transmittance scan (wave_incl_scan_trans) -> T and weight w.  Samples whose
//                         contribution is exactly zero (outside the box, sigma == 0 under ReLU, T == 0)
//                         are dropped; the survivors are compacted into an LDS work list.
//   P1  lanes = channels  LPS lanes cooperate on one sample: each lane fetches 4 consecutive feature
//                         channels (16 B) of each of the 8 corners, so one corner is a contiguous 108-112 B
//                         read.  Interpolate, apply the per-ray SH basis, butterfly-reduce to raw RGB.
//   P2  lanes = samples   sigmoid, accumulate colour / acc / depth, optionally store the per-sample cache.
//
// Numerics follow the reference's CPU path operation by operation where it is cheap (this file is built
// with -ffp-contract=off: every multiply and add below rounds separately unless fmaf is written):
// the interpolation is the ATen grid_sample(align_corners=False, zeros padding) recipe restated in
// oracle/relu_field_oracle.py:trilinear_recipe.
//
// Reference behaviour being replaced (paths relative to the reference repo):
//   thre3d_atom/rendering/volumetric/sample.py, process.py, accumulate.py, utils/spherical_harmonics.py,
//   utils/misc.py:12-50, thre3d_atom/thre3d_reprs/voxels.py:214-331, thre3d_reprs/renderers.py:48-102.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "relu_field.h"

namespace {

// render_emit_direct_kernel: chunks of a ray in flight together / waves per SIMD asked of the register allocator.  Measured on the
// bench step (specular + diffuse launch): 4 chunks at 4 waves 0.056 + 0.055 ms, 3 at 5 0.059 + 0.053, 2 at 6 (76-80 registers, no
// spills) 0.052 + 0.046, 2 at 7 (spills) 0.052 + 0.058, 1 at 8 0.056 + 0.045: occupancy hides the chunk's dependent chain
// (cache load -> key -> cursor atomic -> stores) better than unrolling it does.
#ifndef RF_FWD_WAVES
#define RF_FWD_WAVES 4  // waves per SIMD the forward kernels are register-budgeted for (128 registers)
#endif
#ifndef RF_EMIT_WAVES
#define RF_EMIT_WAVES 6
#endif
#ifndef RF_EMIT_G
#define RF_EMIT_G 2
#endif
constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr float kInfinity = 1e10f;   // thre3d_atom/utils/constants.py:8
constexpr float kZeroPlus = 1e-10f;  // thre3d_atom/utils/constants.py:7

// real SH constants (utils/spherical_harmonics.py:33-52), rounded to float32 like torch does when a
// Python float multiplies a float32 tensor
constexpr float kC0 = 0.28209479177387814f;
constexpr float kC1 = 0.4886025119029199f;
constexpr float kC2_0 = 1.0925484305920792f;
constexpr float kC2_1 = -1.0925484305920792f;
constexpr float kC2_2 = 0.31539156525252005f;
constexpr float kC2_3 = -1.0925484305920792f;
constexpr float kC2_4 = 0.5462742152960396f;
constexpr float kC3_0 = -0.5900435899266435f;
constexpr float kC3_1 = 2.890611442640554f;
constexpr float kC3_2 = -0.4570457994644658f;
constexpr float kC3_3 = 0.3731763325901154f;
constexpr float kC3_4 = -0.4570457994644658f;
constexpr float kC3_5 = 1.445305721320277f;
constexpr float kC3_6 = -0.5900435899266435f;

struct GridArgs {
  const float* dens;
  const float* feat;
  const uint32_t* occ;
  int X, Y, Z, F;
  long long dstride, fstride;
  float amin[3], amax[3], nscale[3], nbias[3];
  float rho;
  int mode;
  int layout;  // channel arrangement. RF_LAYOUT_REFERENCE: dens[.,1] + feat[.,F];  RF_LAYOUT_SPLIT: dens = base[.,4], feat = rest[.,F-3]
  // node order: linear (x, y, z) with z fastest, or (RF_LAYOUT_BRICKED) brick-major: 8^3-node bricks stored contiguously,
  // brick (bx, by, bz) at ((bx * nby + by) * nbz + bz) * 512, node (x & 7, y & 7, z & 7) inside it with z fastest
  int bricked;
  int nby, nbz;
  unsigned int step[3];  // index step to the next node along x, y, z (inside a brick for the bricked order)
  unsigned int jump[3];  // bricked: step from the last node of a brick to the first node of the next brick on that axis
  // "near" addressing (host-checked: <= 2^24 nodes, byte strides < 2^24, both tensors inside one 4 GB window that starts at `base`):
  // a corner's address = uniform 64-bit base (scalar registers) + 32-bit byte offset built with full-rate 24-bit multiplies, instead
  // of a 64-bit multiply-add per corner -- the gathers of the render kernels are bound by vector-ALU issue, not by memory
  const char* base;
  unsigned int dens_off, feat_off;  // byte offsets of dens / feat from base
  int near32;
};

// index of node (x, y, z) in the grid tensors (to be multiplied by the tensor's channel stride)
__device__ __forceinline__ unsigned int node_lin(const GridArgs& g, int x, int y, int z) {
  if (g.bricked)
    return (__umul24(__umul24((unsigned)(x >> 3), (unsigned)g.nby) + (unsigned)(y >> 3), (unsigned)g.nbz) + (unsigned)(z >> 3)) * 512u +
           (unsigned)(((x & 7) << 6) | ((y & 7) << 3) | (z & 7));
  return __umul24(__umul24((unsigned)x, (unsigned)g.Y) + (unsigned)y, (unsigned)g.Z) + (unsigned)z;
}

struct RayArgs {
  const float* origins;
  const float* directions;
  long long n;
  int S;
  float near, far;
  const float* tvals;
  const float* trand;
  // keyed jitter (RF_FLAG_JITTER_KEYED, trand == NULL): u(ray, sample) = hash of (key, ray0 + ray, sample) -- no [N,S] tensor
  unsigned long long jkey;
  long long ray0;  // global index of ray 0 of this batch (jitter stream; pixel index when the camera generates the rays)
  int jitter;
  // rays generated in-kernel from a pinhole camera (origins == NULL): ray r = pixel ray0 + r, row-major
  int cam;
  int H, W;
  float focal;
  float pose[12];  // [3,4] = rotation | translation (camera-to-world)
};

struct OutArgs {
  float* colour;
  float* depth;
  float* acc;
  float* disparity;
  // per-sample cache of the samples that can carry gradient (`need`: inside, T != 0, sigma != 0 under ReLU), COMPACTED per chunk of
  // 64 samples: the j-th such sample of chunk c of ray r sits at slot r * S + 64 c + j, and bit l of cmask[r][c] says that sample
  // 64 c + l is one of them -- the adjoint kernels read nothing else (every other sample contributes exactly nothing)
  float* cache;                // [N,S,4] (raw r, g, b, sigma)
  float* tcache;               // [N,S] transmittance
  unsigned long long* cmask;   // [N, ceil(S / 64)]
  int* stop;                   // [N]
  // binned backward, fused binning: the forward pass counts, per (brick, flags) key, the samples that will emit a record
  int* hist;        // [8 * nbricks] or NULL
  int brick_shift;  // log2 of the brick edge (nodes)
  int nby, nbz;     // bricks along y and z
};

struct GradArgs {
  const float* gcolour;
  const float* gdepth;
  const float* gacc;
  float* gdens;
  float* gfeat;
  // emit mode (binned backward): per-sample records instead of a scatter
  short* keys;      // [N*S] brick id of the sample's cell, kNoBrick for samples without gradient
  float* records;   // [N*S, 4 * record_quads] the record of every keyed slot (formats: record_quads), in slot order
  int brick_shift;  // log2 of the brick edge (nodes)
  int nby, nbz;     // bricks along y and z
  int* hist;        // [8 * nbricks] records per key, added to (may be NULL)
  // direct mode (EMIT == 2): expanded records written straight to their final position
  int* cursor;      // [8 * nbricks] next free position per key
  float4* sorted;   // [capacity, record_quads(K)]
  int* hist_clear;  // counters of the forward pass to clear for the next iteration (may be NULL)
  int nkeys;
};

constexpr short kNoBrick = -1;  // sorts in front of every (brick, flags) key



// 16-byte load/store at 4-byte alignment (a corner's 27 features start at a multiple of 108 B)
struct __attribute__((packed, aligned(4))) f4u {
  float v[4];
};
struct __attribute__((packed, aligned(4))) f3u {
  float v[3];
};
typedef float vf2 __attribute__((ext_vector_type(2)));  // packed pair: v_pk_mul_f32 / v_pk_add_f32
typedef float vf4 __attribute__((ext_vector_type(4)));  // naturally aligned 16-byte vector (non-temporal builtins)
// 16-byte accesses with / without the streaming hint (experiment switches RF_NT_*: data written once and read once by the next kernel)
template <bool NT>
__device__ __forceinline__ void store_f4(float4* p, float4 v) {
  if constexpr (NT) {
    const vf4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(p));
  } else {
    *p = v;
  }
}
template <bool NT>
__device__ __forceinline__ float4 load_f4(const float4* p) {
  if constexpr (NT) {
    const vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(p));
    return make_float4(t[0], t[1], t[2], t[3]);
  } else {
    return *p;
  }
}
#ifndef RF_NT_RECORD_STORE
#define RF_NT_RECORD_STORE 1  // measured (A/B, same box): emit 0.0785 -> 0.0768 ms
#endif
#ifndef RF_NT_CACHE_STORE
#define RF_NT_CACHE_STORE 1   // with the load below: forward pair 0.1955 -> 0.192 ms
#endif
#ifndef RF_NT_RECORD_LOAD
#define RF_NT_RECORD_LOAD 0   // (streaming record loads in the brick pass: 0.331 -> 0.349 ms -- slower)
#endif
#ifndef RF_NT_CACHE_LOAD
#define RF_NT_CACHE_LOAD 1
#endif

// ---------------------------------------------------------------------------------------------
// wave-level helpers (64 lanes)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_fence() {
  // LDS traffic of one wave is executed in order; this only stops the compiler from moving LDS accesses
  // across the phase boundary.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP cross-lane move: lanes without a source (or masked out by ROW_MASK) keep `old`
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                               CTRL, ROW_MASK, 0xf, false));
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143, kDppWaveShr1 = 0x138, kDppWaveShl1 = 0x130;

// Inclusive prefix of a chunk's transmittance factors E_j = 1 - alpha_j, carried in BOTH forms: e = prod E_j and a = 1 - prod E_j (joined
// as a' = a_prefix + a e_prefix).  The product alone is as accurate as torch.cumprod's sequential one in the worst case, but not on
// the rays that matter at high sample counts: with a slowly varying density the factors of neighbouring samples are (nearly) the SAME
// float, the first doubling step rounds all of them the same way and the later steps multiply that error by 32 per chunk -- coherent
// over a ray, where the sequential product's roundings are not: 1000 samples of alpha ~ 1e-4 left the transmittance 1e-5 low (accumulated
// weight of a ray that ends inside the volume 0.99999 instead of 1, depth 6e-5 off: tests/parity_fuzz.py kind "long"; the float32
// reference is within 1e-6 of the float64 value there).  In the `a` form the same roundings are relative to a ~ 1e-4 .. 1e-2, not to 1.
// prefix_transmittance() takes 1 - a while a < 1/4 and the product beyond (exact zeros behind an opaque sample stay exact; a ray
// passes through few chunks in that regime).  DPP (gfx9): 4 row_shr steps inside each row of 16 lanes, then row_bcast:15 / :31 to fold
// the rows; 24 VALU instructions per chunk (the product alone took 6), no LDS traffic.
__device__ __forceinline__ void wave_incl_scan_trans(float& e, float& a) {
  auto step = [&](auto ctrl_tag, auto mask_tag) {
    constexpr int CTRL = decltype(ctrl_tag)::value, MASK = decltype(mask_tag)::value;
    const float ep = dpp_move<CTRL, MASK>(1.0f, e), ap = dpp_move<CTRL, MASK>(0.0f, a);
    a = __builtin_fmaf(a, ep, ap);
    e = e * ep;
  };
  step(std::integral_constant<int, kDppRowShr1>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, kDppRowShr2>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, kDppRowShr4>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, kDppRowShr8>{}, std::integral_constant<int, 0xf>{});
  step(std::integral_constant<int, kDppRowBcast15>{}, std::integral_constant<int, 0xa>{});
  step(std::integral_constant<int, kDppRowBcast31>{}, std::integral_constant<int, 0xc>{});
}
__device__ __forceinline__ float prefix_transmittance(float e, float a) { return (a < 0.25f) ? 1.0f - a : e; }

__device__ __forceinline__ float read_lane(float x, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), lane));
}

// inclusive SUFFIX sum: result[lane] = sum_{l >= lane} x[l]
__device__ __forceinline__ float wave_incl_rscan_add(float x, int lane) {
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    float y = __shfl_down(x, off, kWave);
    if (lane + off < kWave) x = x + y;
  }
  return x;
}

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
  return x;
}

// exp / sigmoid on the hardware transcendental units (v_exp_f32 is 2^x, v_rcp_f32; ~1 ulp each).  The reference's
// own CPU (SLEEF) and GPU (libdevice) paths differ from each other by as much; the parity bar is 1e-5.
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + exp_fast(-x)); }
// alpha = 1 - exp(-x), x = sigma * delta >= 0 (density2occupancy_pb, accumulate.py:24-28), the way the REFERENCE's float32 arithmetic
// produces it: E = exp(-x) correctly rounded, alpha = 1 - E (an exact subtraction: alpha is a multiple of 2^-24 and 1 - alpha == E, so
// that weights alpha_i T_i and transmittances T_i E_i telescope without a bias).  v_exp_f32 is a one-ulp exponential: at the small x of
// densely sampled rays (1000+ samples: x ~ 1e-3) its errors in E near 1 showed as 3..8e-5 on depth at 4000 .. 5000 samples per ray, where the
// float32 reference stays within 2e-6 of the float64 value (tests/parity_fuzz.py, kind "long").  Below 1/16: E = 1 - (x - x^2/2 + ... ),
// the series to 2e-9 relative, i.e. the correctly rounded exponential in all but a handful of cases.  (alpha taken from the series
// itself is MORE accurate than the reference's -- and then 1 - alpha is rounded, with the same sign sample after sample where the
// density varies slowly: accumulated weights of opaque rays came out 1.5e-5 short of 1.  Measured, not kept.)
__device__ __forceinline__ float occupancy_alpha(float x, float& E) {
  float t = __builtin_fmaf(-x, 1.0f / 120.0f, 1.0f / 24.0f);
  t = __builtin_fmaf(-x, t, 1.0f / 6.0f);
  t = __builtin_fmaf(-x, t, 0.5f);
  t = __builtin_fmaf(-x, t, 1.0f);
  const float big = exp_fast(-x);
  E = (x < 0.0625f) ? __builtin_fmaf(-x, t, 1.0f) : big;
  return 1.0f - E;
}
// softplus'(x) = sigmoid(x) = 1 - exp(-softplus(x)), from the cached activated density s = softplus(x) >= 0, to RELATIVE accuracy: a
// sample far below the surface has s ~ 1e-9 and 1 - exp(-s) cancels to 0 in float32, but the last sample of a ray carries
// delta = 1e10 |d| (accumulate.py:49-52), so its d alpha / d sigma is ~1e10 and the product is an ordinary gradient that the reference
// (ATen's softplus backward: z / (z + 1), z = exp(x)) gets right.  Below 0.25 the alternating series, six terms (next term / s < 5e-8).
__device__ __forceinline__ float softplus_slope(float s) {
  if (s < 0.25f) return s * (1.0f - s * (0.5f - s * (1.0f / 6.0f - s * (1.0f / 24.0f - s * (1.0f / 120.0f - s * (1.0f / 720.0f))))));
  return 1.0f - exp_fast(-s);
}

// ---------------------------------------------------------------------------------------------
// sampling (rendering/volumetric/sample.py:39-68)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float z_uniform(float near, float far, float t) { return near * (1.0f - t) + far * t; }

// jitter u in [0, 1) of sample s, or no jitter at all (have_u = false)
__device__ __forceinline__ float z_sample(const float* __restrict__ tv, bool have_u, float u, int s, int S, float near, float far) {
  const float zc = z_uniform(near, far, tv[s]);
  if (!have_u) return zc;
  // stratified jitter: lower/upper = neighbouring mid-points (sample.py:57-64).  (The neighbours are loaded unconditionally, with
  // clamped indices: a load under a lane condition is merged with a copy, and the copy waits for the load on the spot.)
  const float zm = z_uniform(near, far, tv[s > 0 ? s - 1 : 0]), zp = z_uniform(near, far, tv[s < S - 1 ? s + 1 : S - 1]);
  const float lo = (s > 0) ? 0.5f * (zc + zm) : zc;
  const float hi = (s < S - 1) ? 0.5f * (zp + zc) : zc;
  return lo + (hi - lo) * u;
}

// slab test of sample.py:71-184; returns true when the ray hits, writes the (clamped) bounds either way
__device__ __forceinline__ bool ray_box(const float o[3], const float d[3], const float bmin[3], const float bmax[3],
                                        float near, float far, float& t0, float& t1) {
  float lo_run = 0.f, hi_run = 0.f;
  bool hit = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float den = d[a] + kZeroPlus;
    const float ta = (bmin[a] - o[a]) / den;
    const float tb = (bmax[a] - o[a]) / den;
    const bool swap = ta > tb;
    const float lo = swap ? tb : ta;
    const float hi = swap ? ta : tb;
    if (a == 0) {
      lo_run = lo;
      hi_run = hi;
    } else {
      hit = hit && !((lo_run > hi) || (lo > hi_run));
      lo_run = (lo > lo_run) ? lo : lo_run;
      hi_run = (hi < hi_run) ? hi : hi_run;
    }
  }
  t0 = hit ? lo_run : near;
  t1 = hit ? hi_run : far;
  t0 = (t0 < 0.f) ? 0.f : t0;  // torch.clip(min=0): NaN stays NaN, like the reference
  t1 = (t1 < 0.f) ? 0.f : t1;
  return hit;
}

// ---------------------------------------------------------------------------------------------
// per-sample geometry (voxels.py:214-223 normalise, GridSampler.h:27-36 un-normalise)
// ---------------------------------------------------------------------------------------------
struct Cell {
  float idx[3];  // continuous index ((q + 1) * size - 1) / 2
  int i0[3];     // floor of the continuous index (may be -1)
  float w0[3];   // weight of the lower node along each axis  = (i0 + 1) - idx
  float w1[3];   // weight of the upper node                   = idx - i0
};

__device__ __forceinline__ Cell locate(const float p[3], const GridArgs& g) {
  Cell c;
  const int dims[3] = {g.X, g.Y, g.Z};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float q = p[a] * g.nscale[a] + g.nbias[a];
    const float idx = ((q + 1.0f) * (float)dims[a] - 1.0f) / 2.0f;
    const float fl = floorf(idx);
    c.idx[a] = idx;
    c.i0[a] = (int)fl;
    c.w1[a] = idx - fl;
    c.w0[a] = (fl + 1.0f) - idx;
  }
  return c;
}

// LDS work-list entries, 12 dwords each.
// forward:  [0..1] linear voxel index of the (clamped) corner 000   [2] sample lane | step bits << 8
//           [3] unused   [4..11] the 8 trilinear corner weights (0 for corners outside the grid)
//           step bits: bit0/1/2 = the x/y/z upper node is a distinct voxel (clamping collapses it at the border)
// backward: [0] (ix0+1) | (iy0+1) << 11 | (iz0+1) << 22 (grid dims <= 2046)   [1] sample lane
//           [2..7] w0x w1x w0y w1y w0z w1z   [8..11] dL/d(pre-activation density), dL/d(raw r, g, b)
constexpr int kEntryFwd = 12;
constexpr int kEntryBwd = 12;

__device__ __forceinline__ uint32_t pack_cell(const Cell& c) {
  return (uint32_t)(c.i0[0] + 1) | ((uint32_t)(c.i0[1] + 1) << 11) | ((uint32_t)(c.i0[2] + 1) << 22);
}

struct Corners {
  unsigned int lin[8];  // linear voxel index of the (clamped) corner k = dx + 2 dy + 4 dz
  unsigned int s[3];    // index steps from corner 000 to its x / y / z upper neighbour (0 where clamping collapses them)
  float w[8];           // trilinear weight, forced to 0 for corners outside the grid
};

// grid dims <= 2046, so every product below has 24-bit operands (v_mul_u32_u24 runs at full rate, v_mul_lo_u32 does
// not); voxel counts are limited to 2^32 / stride by the host-side check.
__device__ __forceinline__ Corners corners_of(uint32_t packed, const float wts[6], const GridArgs& g) {
  const int ix0 = (int)(packed & 0x7ffu) - 1, iy0 = (int)((packed >> 11) & 0x7ffu) - 1, iz0 = (int)(packed >> 22) - 1;
  const bool okx[2] = {ix0 >= 0, ix0 + 1 < g.X};  // the point is inside the box: ix0 in [-1, X-1]
  const bool oky[2] = {iy0 >= 0, iy0 + 1 < g.Y};
  const bool okz[2] = {iz0 >= 0, iz0 + 1 < g.Z};
  const int cx0 = max(ix0, 0), cy0 = max(iy0, 0), cz0 = max(iz0, 0);
  const unsigned int lin0 = node_lin(g, cx0, cy0, cz0);
  // step to the upper node: a whole voxel (a jump into the next brick when the lower node is a brick's last one), or 0
  // when the clamped upper node coincides with the lower one
  const unsigned int sx = (okx[0] && okx[1]) ? ((g.bricked && (cx0 & 7) == 7) ? g.jump[0] : g.step[0]) : 0u;
  const unsigned int sy = (oky[0] && oky[1]) ? ((g.bricked && (cy0 & 7) == 7) ? g.jump[1] : g.step[1]) : 0u;
  const unsigned int sz = (okz[0] && okz[1]) ? ((g.bricked && (cz0 & 7) == 7) ? g.jump[2] : g.step[2]) : 0u;
  // (a node outside the grid gets weight 0: masked per axis -- 0 times the finite weights of the other axes is the same +0)
  const float ax[2] = {okx[0] ? wts[0] : 0.0f, okx[1] ? wts[1] : 0.0f};
  const float ay[2] = {oky[0] ? wts[2] : 0.0f, oky[1] ? wts[3] : 0.0f};
  const float az[2] = {okz[0] ? wts[4] : 0.0f, okz[1] ? wts[5] : 0.0f};
  const float wxy[4] = {ax[0] * ay[0], ax[1] * ay[0], ax[0] * ay[1], ax[1] * ay[1]};  // [dx + 2 dy]
  Corners c;
  c.s[0] = sx;
  c.s[1] = sy;
  c.s[2] = sz;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
    c.lin[k] = lin0 + (dx ? sx : 0u) + (dy ? sy : 0u) + (dz ? sz : 0u);
    c.w[k] = wxy[dx + 2 * dy] * az[dz];
  }
  return c;
}

__device__ __forceinline__ float density_post(float pre, int mode);

// density: post( interp( pre(D * rho) ) )  (voxels.py:292-309)
__device__ __forceinline__ float interp_density(const Corners& c, const GridArgs& g, float& pre_out) {
  float raw[8];
  if (g.near32) {  // (wave-uniform)
    const unsigned int sb = (unsigned int)g.dstride * 4u;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned int o = __umul24(c.lin[k], sb);
      raw[k] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(g.dens) + (size_t)o);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) raw[k] = g.dens[c.lin[k] * g.dstride];
    asm volatile("" ::: "memory");  // (keeps the two branches' loads apart: merged, they lose the scalar-base addressing)
  }
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float v = raw[k] * g.rho;
    if (g.mode == RF_DENSITY_ABS) v = fabsf(v);
    acc = acc + v * c.w[k];
  }
  pre_out = acc;
  if (g.mode == RF_DENSITY_RELU) return fmaxf(acc, 0.0f);
  if (g.mode == RF_DENSITY_SOFTPLUS) return (acc > 20.0f) ? acc : log1pf(expf(acc));
  return acc;
}

// signed SH basis in the reference's operation order (utils/spherical_harmonics.py:86-116)
template <int K>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float Y[16]) {
  Y[0] = kC0;
  if (K > 1) {
    Y[1] = -(kC1 * y);
    Y[2] = kC1 * z;
    Y[3] = -(kC1 * x);
  }
  if (K > 4) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    Y[4] = kC2_0 * xy;
    Y[5] = kC2_1 * yz;
    Y[6] = kC2_2 * ((2.0f * zz - xx) - yy);
    Y[7] = kC2_3 * xz;
    Y[8] = kC2_4 * (xx - yy);
    if (K > 9) {
      Y[9] = (kC3_0 * y) * (3.0f * xx - yy);
      Y[10] = (kC3_1 * xy) * z;
      Y[11] = (kC3_2 * y) * ((4.0f * zz - xx) - yy);
      Y[12] = (kC3_3 * z) * ((2.0f * zz - 3.0f * xx) - 3.0f * yy);
      Y[13] = (kC3_4 * x) * ((4.0f * zz - xx) - yy);
      Y[14] = (kC3_5 * z) * (xx - yy);
      Y[15] = (kC3_6 * x) * (xx - 3.0f * yy);
    }
  }
}

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

// Keyed jitter of the stratified sampler (sample.py:57-64 draws torch.rand(N, S)): a counter-based generator instead of a
// [N, S] tensor.  u = 24 random bits of mix32 over (per-ray seed, sample index); the per-ray seed hashes the 64-bit key with
// the global ray index.  Restated in oracle/relu_field_oracle.py:keyed_jitter (tests compare the two bit for bit).
__device__ __forceinline__ uint32_t jitter_ray_seed(unsigned long long key, long long gray) {
  const uint32_t a = mix32((uint32_t)gray ^ (uint32_t)key);
  const uint32_t b = mix32((uint32_t)((unsigned long long)gray >> 32) + (uint32_t)(key >> 32));
  return a ^ (b * 0x9E3779B9u + 0x85EBCA6Bu);
}
__device__ __forceinline__ float jitter_uniform(uint32_t ray_seed, int s) {
  const uint32_t h = mix32(ray_seed + (uint32_t)s * 0x9E3779B9u);
  return (float)(h >> 8) * 5.9604644775390625e-08f;  // [0, 1) on the 2^-24 lattice
}

__device__ __forceinline__ void pixel_ray(int i, int j, int H, int W, float focal, const float* R, float d[3]) {
  // pixel centres; camera looks along -z, y up
  const float cx = (((float)j + 0.5f) - (float)W * 0.5f) / focal;
  const float cy = -((((float)i + 0.5f) - (float)H * 0.5f) / focal);
  const float cz = -1.0f;
#pragma unroll
  for (int a = 0; a < 3; ++a) d[a] = (R[a * 3 + 0] * cx + R[a * 3 + 1] * cy) + R[a * 3 + 2] * cz;
}

// ---------------------------------------------------------------------------------------------
// per-ray state shared by forward and backward
// ---------------------------------------------------------------------------------------------
struct RayState {
  float o[3], d[3];
  float dnorm;
  float near, far;
  uint32_t jseed;  // keyed jitter: per-ray seed
};

__device__ __forceinline__ RayState load_ray(const RayArgs& r, const GridArgs& g, long long ray, uint32_t flags) {
  RayState st;
  if (r.cam) {  // cast_rays (utils/misc.py:12-50) fused: pixel p = ray0 + ray, row-major
    const long long pidx = r.ray0 + ray;
    const int i = (int)(pidx / r.W), j = (int)(pidx - (long long)i * r.W);
    const float R[9] = {r.pose[0], r.pose[1], r.pose[2], r.pose[4], r.pose[5], r.pose[6], r.pose[8], r.pose[9], r.pose[10]};
    pixel_ray(i, j, r.H, r.W, r.focal, R, st.d);
    st.o[0] = r.pose[3];
    st.o[1] = r.pose[7];
    st.o[2] = r.pose[11];
  } else {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      st.o[a] = r.origins[ray * 3 + a];
      st.d[a] = r.directions[ray * 3 + a];
    }
  }
  st.jseed = r.jitter ? jitter_ray_seed(r.jkey, r.ray0 + ray) : 0u;
  st.dnorm = sqrtf((st.d[0] * st.d[0] + st.d[1] * st.d[1]) + st.d[2] * st.d[2]);
  st.near = r.near;
  st.far = r.far;
  if (flags & RF_FLAG_AABB_SAMPLING) {
    float t0, t1;
    ray_box(st.o, st.d, g.amin, g.amax, r.near, r.far, t0, t1);
    st.near = t0;
    st.far = t1;
  }
  return st;
}

// everything P0 knows about one sample
struct Sample {
  float z, delta;
  bool valid, inside;
  Cell cell;
};

// the sample of lane-index s given its ray parameter z and the next sample's parameter z_next
__device__ __forceinline__ Sample sample_at(const RayState& st, const RayArgs& r, const GridArgs& g, int s, float z,
                                            float z_next) {
  Sample sm;
  sm.valid = s < r.S;
  sm.z = z;
  sm.delta = (s < r.S - 1) ? (z_next - z) * st.dnorm : kInfinity * st.dnorm;  // accumulate.py:50-55
  float p[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) p[a] = st.o[a] + st.d[a] * sm.z;
  sm.inside = sm.valid && p[0] > g.amin[0] && p[0] < g.amax[0] && p[1] > g.amin[1] && p[1] < g.amax[1] &&
              p[2] > g.amin[2] && p[2] < g.amax[2];  // strict, voxels.py:252-274
  sm.cell = locate(p, g);
  return sm;
}

__device__ __forceinline__ float z_of(const RayState& st, const RayArgs& r, long long ray, int s) {
  const int sc = min(s, r.S - 1);
  float u = 0.0f;
  if (r.trand)
    u = r.trand[ray * (long long)r.S + sc];
  else if (r.jitter)
    u = jitter_uniform(st.jseed, sc);
  return z_sample(r.tvals, r.trand != nullptr || r.jitter, u, sc, r.S, st.near, st.far);
}

// z_of in two halves, for kernels that want every load of several samples in flight before the first one is used:
// z_requests issues the (unconditional, clamped) t_vals reads, z_from does z_of's arithmetic on them -- bit for bit the same value.
struct ZRequests {
  float t0, tm, tp;
};
__device__ __forceinline__ ZRequests z_requests(const RayArgs& r, int s) {
  const int sc = min(s, r.S - 1);
  ZRequests q;
  q.t0 = r.tvals[sc];
  q.tm = r.tvals[sc > 0 ? sc - 1 : 0];
  q.tp = r.tvals[sc < r.S - 1 ? sc + 1 : r.S - 1];
  return q;
}
// (a jitter TABLE, if one is used instead of the keyed generator, is read here, synchronously)
__device__ __forceinline__ float z_from(const RayState& st, const RayArgs& r, const ZRequests& q, long long ray, int s) {
  const int sc = min(s, r.S - 1);
  const float zc = z_uniform(st.near, st.far, q.t0);
  if (!(r.trand != nullptr || r.jitter)) return zc;
  const float u = r.trand ? r.trand[ray * (long long)r.S + sc] : jitter_uniform(st.jseed, sc);
  const float zm = z_uniform(st.near, st.far, q.tm), zp = z_uniform(st.near, st.far, q.tp);
  const float lo = (sc > 0) ? 0.5f * (zc + zm) : zc;
  const float hi = (sc < r.S - 1) ? 0.5f * (zp + zc) : zc;
  return lo + (hi - lo) * u;
}

__device__ __forceinline__ Sample make_sample(const RayState& st, const RayArgs& r, const GridArgs& g, long long ray,
                                              int s) {
  return sample_at(st, r, g, s, z_of(st, r, ray, s), z_of(st, r, ray, s + 1));
}

__device__ __forceinline__ bool cell_occupied(const Cell& c, const GridArgs& g) {
  // bit index over the (X+1)(Y+1)(Z+1) cells whose lower node is i0 in [-1, dim-1]
  const long long bit = ((long long)(c.i0[0] + 1) * (g.Y + 1) + (c.i0[1] + 1)) * (g.Z + 1) + (c.i0[2] + 1);
  return (g.occ[bit >> 5] >> (bit & 31)) & 1u;
}

// ---------------------------------------------------------------------------------------------
// feature gather for LPS lanes per sample (P1).  Returns raw RGB in every lane of the group.
//   CORNER mode (K_READ == 1; SH degree 0 or render_diffuse): lane = corner, 3 channels per lane.
//   CHANNEL mode: each lane gathers 4 consecutive floats per corner (see LaneSlice).
// ---------------------------------------------------------------------------------------------
template <int K, bool DIFFUSE>
struct Layout {
  static constexpr bool kCorner = DIFFUSE || K == 1;
  static constexpr int kF = 3 * K;
  static constexpr int kLPS = kCorner ? 8 : (K == 4 ? 4 : (K == 9 ? 8 : 16));
  static constexpr int kGroups = kWave / kLPS;
};

// sum over each aligned group of LPS lanes (LPS <= 16, inside one DPP row); the total lands in the LAST lane of the
// group -- row_shr adds fuse into v_add_f32_dpp, one instruction per step
template <int LPS>
__device__ __forceinline__ float group_sum_to_last(float x) {
  if (LPS >= 16) x += dpp_move<kDppRowShr8, 0xf>(0.0f, x);
  if (LPS >= 8) x += dpp_move<kDppRowShr4, 0xf>(0.0f, x);
  if (LPS >= 4) x += dpp_move<kDppRowShr2, 0xf>(0.0f, x);
  x += dpp_move<kDppRowShr1, 0xf>(0.0f, x);
  return x;
}


// What one P1 lane gathers: 4 consecutive floats at `src + voxel * stride + off` of every corner, and for each of
// the 4 elements which colour it feeds (chan, -1 = not owned) with which SH weight (yb).
//   reference layout: lane `sub` owns features [off, off+4), off = min(4 sub, F-4): the last lane is shifted back so
//     that it never reads past the corner's F features; its first j0 elements belong to the previous lane.
//   split layout: lane 0 reads the 16-byte base record (sigma, sh0 r, g, b); lanes 1.. read the rest record
//     (index = colour * (K-1) + (k-1)) four floats at a time, last lane shifted back likewise.
// The per-ray basis is staged through a 16-float LDS row so that the dynamic index never touches scratch.
struct LaneSlice {
  const float* src;
  long long stride;
  int off;
  bool active;
  float yb[4];
  int chan[4];
  int colour;               // split layout: the one colour all owned elements of this lane feed (-1: base lane, inactive lane)
  bool closes_colour;       // split layout: last lane of its colour's run of lanes
  unsigned int strideB, offB;  // near addressing: byte stride of the lane's tensor, byte offset of the lane's slice from g.base
};

template <int K, int LPS>
__device__ __forceinline__ LaneSlice lane_slice(const GridArgs& g, const float d[3], float dnorm, int lane,
                                                float* ldsY) {
  constexpr int F = 3 * K;
  constexpr int R = 3 * (K - 1);
  float Y[16];
  sh_basis<K>(d[0] / dnorm, d[1] / dnorm, d[2] / dnorm, Y);  // v = d / |d| (process.py:53)
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) ldsY[k] = Y[k];
  }
  wave_lds_fence();
  const int sub = lane % LPS;
  LaneSlice ls;
  if (g.layout == RF_LAYOUT_SPLIT) {
    if (sub == 0) {
      // (idle: P0 reads the whole 16-byte base record of a corner for the density and interpolates the degree-0 colour there)
      ls.src = g.dens;
      ls.stride = g.dstride;
      ls.off = 0;
      ls.active = false;
      ls.colour = -1;
      ls.closes_colour = false;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ls.chan[j] = -1;
        ls.yb[j] = 0.0f;
      }
    } else {
      // colour-aligned: lanes 1 + c * LC + q, q < LC = ceil((K-1) / 4), hold the rest coefficients 4q.. of colour c, so that a
      // lane's four products feed ONE colour and the reduction is a sum over LC neighbouring lanes (rest_colour_sums)
      constexpr int KR = K - 1, LC = (KR + 3) / 4;
      const int t = sub - 1;
      const int c = t / LC, q = t - c * LC;
      const int first = c * KR + 4 * q;  // first owned rest element
      const int last = min(first + 4, (c + 1) * KR);
      ls.src = g.feat;
      ls.stride = g.fstride;
      ls.off = min(first, R - 4);  // (the read never leaves the node's record)
      ls.active = c < 3;
      ls.colour = ls.active ? c : -1;
      ls.closes_colour = ls.active && q == LC - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rr = ls.off + j;
        const bool own = ls.active && rr >= first && rr < last;
        ls.chan[j] = own ? c : -1;
        ls.yb[j] = own ? ldsY[rr - c * KR + 1] : 0.0f;
      }
    }
  } else {
    ls.colour = -1;
    ls.closes_colour = false;
    ls.src = g.feat;
    ls.stride = g.fstride;
    ls.off = min(4 * sub, F - 4);
    ls.active = 4 * sub < F;
    const int j0 = 4 * sub - ls.off;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = ls.off + j;
      const bool own = ls.active && (j >= j0);
      ls.chan[j] = own ? f / K : -1;
      ls.yb[j] = own ? ldsY[f % K] : 0.0f;
    }
  }
  ls.strideB = (unsigned int)ls.stride * 4u;
  ls.offB = (ls.src == g.dens ? g.dens_off : g.feat_off) + (unsigned int)ls.off * 4u;
  wave_lds_fence();
  return ls;
}

// Split layout, colour-aligned lanes (lane_slice): p[j] = basis * interpolated coefficient of the lane's four elements.  Returns, in
// the lane that closes colour c (LaneSlice::closes_colour), the sum of the colour's rest terms (the LC lanes in front of and
// including this one).  Call with all lanes of the wave enabled.
template <int K>
__device__ __forceinline__ float rest_colour_sums(const float p[4]) {
  constexpr int LC = (K - 1 + 3) / 4;
  float dot = (p[0] + p[1]) + (p[2] + p[3]);
  if (LC >= 2) dot += dpp_move<kDppRowShr1, 0xf>(0.0f, dot);
  if (LC >= 4) dot += dpp_move<kDppRowShr2, 0xf>(0.0f, dot);
  return dot;
}

// Lanes hold the keys of 64 consecutive sample slots (-1 = none).  Consecutive samples of a ray mostly share their
// key, so one atomic per RUN of equal keys is issued (by the run's first lane) instead of one per lane.  Returns this
// lane's rank = counter value before the run + position inside the run (only meaningful for key >= 0); with
// WANT_RANK = false the atomic is fire-and-forget.
template <bool WANT_RANK>
__device__ __forceinline__ int add_key_runs(int* counters, int key, int lane) {
  const bool active = key >= 0;
  const int prev = __shfl_up(key, 1);
  const bool head = active && (lane == 0 || prev != key);
  const unsigned long long heads = __ballot(head);
  const unsigned long long ends = __ballot(head || !active);  // a run stops at the next head or inactive lane
  const unsigned long long above = (lane == 63) ? 0ull : (ends & ~((2ull << lane) - 1ull));
  const int run = (above ? __builtin_ctzll(above) : 64) - lane;
  int base = 0;
  if (head) {
    if (WANT_RANK)
      base = atomicAdd(&counters[key], run);
    else
      atomicAdd(&counters[key], run);
  }
  if (!WANT_RANK) return 0;
  const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
  const int hl = below ? 63 - __builtin_clzll(below) : 0;
  return __shfl(base, hl) + (lane - hl);
}

// LDS channel order of a node ("split order"): 0 = density, 1..3 = degree-0 R,G,B, 4 + r = rest channel r
template <int K>
__device__ __forceinline__ void lds_channel_meaning(int c, int& colour, int& basis_k) {
  if (c == 0) {
    colour = 3;
    basis_k = 0;
  } else if (c < 4) {
    colour = c - 1;
    basis_k = 0;
  } else {
    const int rr = c - 4;
    constexpr int KR = K > 1 ? K - 1 : 1;
    colour = (K > 1) ? rr / KR : 0;
    basis_k = (K > 1) ? rr % KR + 1 : 0;
  }
}

// float4s of a gradient record of the binned backward:
//   K == 1 (render_diffuse passes, degree-0 grids): (index x, y, z, -) (d density, d sh0 r, g, b)                       32 B
//   K  > 1: (index x, y, z, d density) (d raw r, g, b, v_x) (v_y, v_z, -, -), v = the ray's unit viewing direction       48 B
//           -- the per-channel values d raw[colour] * Y_k(v) are expanded by the brick pass in LDS, never in HBM
__host__ __device__ constexpr int record_quads(int K) { return K == 1 ? 2 : 3; }
// channel values of a compact record in LDS channel order (lds_channel_meaning): 0 = density, 1..3 = degree 0, 4 + rr = rest rr
template <int K>
__device__ __forceinline__ float record_channel(int ch, float gdens, const float graw[3], const float Y[16]) {
  if (ch == 0) return gdens;
  int colour, basis_k;
  lds_channel_meaning<K>(ch, colour, basis_k);
  return graw[colour] * Y[basis_k];
}

// key of a sample's cell in the binned backward.  Brick (bx, by, bz) = the brick of the cell's lower node; flag f_a = the cell's
// UPPER node on axis a belongs to the next brick (and exists), i.e. the record also touches nodes of that neighbour.
//   key = ((((bx * 2 + f_x) * nby + by) * nbz + bz) << 2) | f_y | f_z << 1
// x-slab major with the x flag directly below the slab index: everything that touches the nodes of the x-slabs [s0, s1) of bricks --
// the records of those slabs plus the x-flagged records of slab s0 - 1 -- is ONE contiguous key range.  That is what lets a
// data-parallel rank send an owner of a slab range its share of a sorted list as a single slice (owner-computes exchange).
// `shift`: log2 of the brick edge; bits 4.. = by how much the x edge's log2 is smaller (brick_geometry: 0 for cubic bricks, 1 for the
// 4 x 8 x 8 bricks of RF_BRICK_4X8X8).
__device__ __forceinline__ int brick_key(const int i0[3], const GridArgs& g, int shift, int nby, int nbz) {
  const int dims3[3] = {g.X, g.Y, g.Z};
  const int syz = shift & 15;
  const int sh3[3] = {syz - (shift >> 4), syz, syz};
  int b3[3], f3[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int lo = max(i0[a], 0), up = i0[a] + 1;
    b3[a] = lo >> sh3[a];
    f3[a] = (up < dims3[a] && (up >> sh3[a]) != b3[a]) ? 1 : 0;
  }
  // (brick counts are far below 2^24: full-rate 24-bit multiplies)
  return (int)((__umul24(__umul24((unsigned)((b3[0] << 1) | f3[0]), (unsigned)nby) + (unsigned)b3[1], (unsigned)nbz) + (unsigned)b3[2]) << 2) | f3[1] | (f3[2] << 1);
}

// Parameter interval of a ray inside the box (same slab test as the AABB sampler, reciprocal-based: only used with a generous
// margin, never for sample positions).  A chunk of 64 samples that lies outside it by that margin contributes exactly
// nothing -- sigma = 0 -> alpha = 0 -> w = 0, T unchanged, no gradient -- and is skipped without evaluating a single sample.
struct BoxSpan {
  float t_in, t_out, zpad, margin;
  bool hits;
};

__device__ __forceinline__ BoxSpan box_span(const RayState& st, const RayArgs& r, const GridArgs& g) {
  BoxSpan b;
  b.t_in = -kInfinity;
  b.t_out = kInfinity;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float inv = __builtin_amdgcn_rcpf(st.d[a] + kZeroPlus);
    const float ta = (g.amin[a] - st.o[a]) * inv, tb = (g.amax[a] - st.o[a]) * inv;
    b.t_in = fmaxf(b.t_in, fminf(ta, tb));
    b.t_out = fminf(b.t_out, fmaxf(ta, tb));
  }
  b.margin = 1e-3f * (1.0f + fabsf(b.t_in) + fabsf(b.t_out));
  b.hits = b.t_out >= b.t_in - b.margin;
  b.zpad = fabsf(st.far - st.near) / (float)(r.S > 1 ? r.S - 1 : 1);  // jitter stays within one stratum
  return b;
}

// wave-uniform: no sample of chunk `chunk` can be inside the box
__device__ __forceinline__ bool chunk_outside_box(const BoxSpan& b, const RayState& st, const RayArgs& r, int chunk) {
  const int s_first = chunk * kWave, s_last = min(r.S - 1, chunk * kWave + kWave - 1);
  const float za = z_uniform(st.near, st.far, r.tvals[s_first]), zb = z_uniform(st.near, st.far, r.tvals[s_last]);
  const float zlo = fminf(za, zb) - b.zpad, zhi = fmaxf(za, zb) + b.zpad;
  return !b.hits || zhi < b.t_in - b.margin || zlo > b.t_out + b.margin;
}

// =============================================================================================
// forward
// =============================================================================================
// one wave = one ray; my_entry / my_rgb: the wave's own kWave * kEntryFwd words and kWave * 4 floats of LDS
template <int K, bool DIFFUSE, bool SAVE>
__device__ __forceinline__ void render_forward_ray(const GridArgs& g, const RayArgs& r, const OutArgs& out, uint32_t flags, long long ray, int lane,
                                                   uint32_t* my_entry, float* my_rgb) {
  using L = Layout<K, DIFFUSE>;
  constexpr int LPS = L::kLPS;
  constexpr int GROUPS = L::kGroups;
  if (ray >= r.n) return;

  const RayState st = load_ray(r, g, ray, flags);
  const bool white = flags & RF_FLAG_WHITE_BKGD;
  const bool use_occ = (flags & RF_FLAG_OCCUPANCY_SKIP) && g.occ != nullptr;

  // per-ray SH basis, then the per-lane slice of it used in P1
  const int sub = lane % LPS;
  const int group = lane / LPS;
  LaneSlice ls;
  ls.src = nullptr;
  ls.active = false;
  if constexpr (!L::kCorner) ls = lane_slice<K, LPS>(g, st.d, st.dnorm, lane, my_rgb);

  float T_carry = 1.0f;
  float part_c[3] = {0.f, 0.f, 0.f};
  float part_acc = 0.f, part_depth = 0.f;
  int processed = 0;

  const BoxSpan span = box_span(st, r, g);

  // z of the current chunk is computed one iteration ahead, so that the last lane can take its "next sample" from
  // lane 0 of the following chunk: one z evaluation per sample instead of two
  float z_cur = 0.0f;
  bool z_ready = false;

  const int nchunks = (r.S + kWave - 1) / kWave;
  // SAVE: lane c keeps the mask of the cached samples of chunk 64 g + c of the current group g of 64 chunks
  unsigned long long my_cmask = 0ull;
  int cmask_group = 0;
  auto flush_cmasks = [&](int group) {
    const int c = group * kWave + lane;
    if (c < nchunks) out.cmask[ray * (long long)nchunks + c] = my_cmask;
    my_cmask = 0ull;
  };
  // Which chunks cannot hold a sample inside the box: lane c decides for chunk c, all table reads in flight at once (a ray
  // of more than 64 chunks falls back to one decision per iteration).  The t_vals reads of a chunk's samples are requested
  // one iteration before they are needed: per chunk these small dependent loads were 2-3 exposed cache latencies in front of
  // the corner gather.
  const unsigned long long empty_mask = __ballot(lane < nchunks && chunk_outside_box(span, st, r, min(lane, nchunks - 1)));
  ZRequests zq_ahead = z_requests(r, lane + kWave);  // samples of chunk 1
  bool zq_valid = true;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int s = chunk * kWave + lane;
    if constexpr (SAVE) {
      if ((chunk >> 6) != cmask_group) {  // (rays of more than 4096 samples)
        flush_cmasks(cmask_group);
        cmask_group = chunk >> 6;
      }
    }
    {
      const bool empty = chunk < kWave ? (bool)((empty_mask >> chunk) & 1ull) : chunk_outside_box(span, st, r, chunk);
      if (empty) {  // wave-uniform
        // (nothing is cached for such a chunk: its mask stays 0)
        processed = min(r.S, (chunk + 1) * kWave);
        z_ready = false;
        zq_valid = false;
        continue;
      }
    }
    // ---------------- P0: lanes = samples ----------------
    if (!z_ready) z_cur = z_of(st, r, ray, s);
    const float z_nxt = zq_valid ? z_from(st, r, zq_ahead, ray, s + kWave) : z_of(st, r, ray, s + kWave);  // the following chunk (clamped to the last sample)
    zq_ahead = z_requests(r, s + 2 * kWave);
    zq_valid = true;
    const float z_up = dpp_move<kDppWaveShl1, 0xf>(0.0f, z_cur);  // lane i <- lane i+1
    const float z_next = (lane == kWave - 1) ? read_lane(z_nxt, 0) : z_up;
    Sample sm = sample_at(st, r, g, s, z_cur, z_next);
    z_cur = z_nxt;
    z_ready = true;
    bool live = sm.inside;
    if (use_occ && live) live = cell_occupied(sm.cell, g);
    float sigma = 0.0f;
    Corners cn;
    float wts[6] = {sm.cell.w0[0], sm.cell.w1[0], sm.cell.w0[1], sm.cell.w1[1], sm.cell.w0[2], sm.cell.w1[2]};
    const uint32_t packed = pack_cell(sm.cell);
    // Split storage: the 16-byte base record of a corner holds the density AND the three degree-0 coefficients, so P0's one gather
    // per corner already brings the degree-0 colour -- interpolated right here (ATen's corner order).  Degree-0 / render_diffuse
    // passes then skip the work list and P1 altogether; the others gather only the rest coefficients in P1.
    const bool base_in_p0 = g.layout == RF_LAYOUT_SPLIT;  // wave-uniform
    const bool fast_base = L::kCorner && base_in_p0;
    float fast_rgb[3] = {0.f, 0.f, 0.f};
    if (live) {
      cn = corners_of(packed, wts, g);
      if (base_in_p0) {
        f4u t[8];
        if (g.near32) {
          const unsigned int sb = (unsigned int)g.dstride * 4u;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const unsigned int o = __umul24(cn.lin[k], sb);
            t[k] = *reinterpret_cast<const f4u*>(reinterpret_cast<const char*>(g.dens) + (size_t)o);
          }
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) t[k] = *reinterpret_cast<const f4u*>(g.dens + cn.lin[k] * g.dstride);
          asm volatile("" ::: "memory");  // (keeps the two branches' loads apart: merged, they lose the scalar-base addressing)
        }
        // density in the reference's operation order (separately rounded products, ATen's corner order: it decides which samples
        // the ReLU keeps); the colour coefficients with fused multiply-adds, two per instruction
        float acc = 0.0f, cb = 0.0f;
        vf2 crg = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float v = t[k].v[0] * g.rho;
          if (g.mode == RF_DENSITY_ABS) v = fabsf(v);
          acc = acc + v * cn.w[k];
          const vf2 rg = {t[k].v[1], t[k].v[2]}, wk = {cn.w[k], cn.w[k]};
          crg = __builtin_elementwise_fma(rg, wk, crg);
          cb = __builtin_fmaf(t[k].v[3], cn.w[k], cb);
        }
        sigma = density_post(acc, g.mode);
        fast_rgb[0] = kC0 * crg.x;
        fast_rgb[1] = kC0 * crg.y;
        fast_rgb[2] = kC0 * cb;
      } else {
        float pre;
        sigma = interp_density(cn, g, pre);
      }
    }
    float one_minus;  // = exp(-sigma delta) = 1 - alpha exactly
    const float alpha = occupancy_alpha(sigma * sm.delta, one_minus);  // density2occupancy_pb, accumulate.py:24-28
    float incl_e = sm.valid ? one_minus : 1.0f, incl_a = sm.valid ? alpha : 0.0f;
    wave_incl_scan_trans(incl_e, incl_a);
    const float incl = prefix_transmittance(incl_e, incl_a);
    const float excl = dpp_move<kDppWaveShr1, 0xf>(1.0f, incl);  // lane i <- lane i-1, lane 0 <- 1
    const float T = T_carry * excl;
    const float w = alpha * T;
    T_carry = T_carry * read_lane(incl, kWave - 1);
    processed = min(r.S, (chunk + 1) * kWave);

    // a sample needs its colour only if it can contribute: inside, T != 0 and (under ReLU) sigma != 0
    const bool need = live && (T != 0.0f) && !(g.mode == RF_DENSITY_RELU && sigma == 0.0f);
    const unsigned long long mask = __ballot(need);
    const int count = fast_base ? 0 : __popcll(mask);
    const int slot = __popcll(mask & ((1ull << lane) - 1ull));
    if (need && !fast_base) {
      uint32_t* e = my_entry + slot * kEntryFwd;
      e[0] = cn.lin[0];
      e[1] = cn.s[0];
      e[2] = cn.s[1];
      e[3] = cn.s[2];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[4 + k] = __float_as_uint(cn.w[k]);
    }
    wave_lds_fence();

    // ---------------- P1: LPS lanes per sample ----------------
    // (the raw colour of the sample in work-list slot j lands in my_rgb[4 j ..]; P2's lane reads the slot it filled)
    for (int base = 0; base < count; base += GROUPS) {
      const int slot = base + group;
      const bool has = slot < count;
      float rgb[3] = {0.f, 0.f, 0.f};
      float prod[4] = {0.f, 0.f, 0.f, 0.f};
      if (has) {
        const uint32_t* e = my_entry + slot * kEntryFwd;
        const unsigned int lin0 = e[0];
        // corner k = dx + 2 dy + 4 dz; the steps to the upper nodes are whole voxels (a jump into the next brick in the bricked
        // node order) or 0 (clamped at the border)
        const unsigned int sx = e[1], sy = e[2], sz = e[3];
        float w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = __uint_as_float(e[4 + k]);
        if constexpr (L::kCorner) {
          // lane = corner `sub`; channels 0, K_full, 2*K_full of that corner
          const int kfull = g.F / 3;
          const unsigned int lin = lin0 + ((sub & 1) ? sx : 0u) + ((sub & 2) ? sy : 0u) + ((sub & 4) ? sz : 0u);
          float wk = w[0];
#pragma unroll
          for (int k = 1; k < 8; ++k) wk = (sub == k) ? w[k] : wk;
          float v0, v1, v2;
          if (g.layout == RF_LAYOUT_SPLIT) {
            // base record = (sigma, sh0 r, sh0 g, sh0 b): one 16-byte load per corner
            const f4u t = *reinterpret_cast<const f4u*>(g.dens + lin * g.dstride);
            v0 = t.v[1];
            v1 = t.v[2];
            v2 = t.v[3];
          } else if (kfull == 1) {
            const float* fp = g.feat + lin * g.fstride;
            const f3u t = *reinterpret_cast<const f3u*>(fp);
            v0 = t.v[0];
            v1 = t.v[1];
            v2 = t.v[2];
          } else {
            const float* fp = g.feat + lin * g.fstride;
            v0 = fp[0];
            v1 = fp[kfull];
            v2 = fp[2 * kfull];
          }
          rgb[0] = v0 * wk;
          rgb[1] = v1 * wk;
          rgb[2] = v2 * wk;
        } else {
          vf2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
          if (ls.active) {
            f4u v[8];
            if (g.near32) {  // (wave-uniform)
              const unsigned int o0 = __umul24(lin0, ls.strideB) + ls.offB;
              const unsigned int bx = __umul24(sx, ls.strideB), by = __umul24(sy, ls.strideB), bz = __umul24(sz, ls.strideB);
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const unsigned int o = o0 + ((k & 1) ? bx : 0u) + ((k & 2) ? by : 0u) + ((k & 4) ? bz : 0u);
                v[k] = *reinterpret_cast<const f4u*>(g.base + (size_t)o);
              }
            } else {
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const unsigned int lin = lin0 + ((k & 1) ? sx : 0u) + ((k & 2) ? sy : 0u) + ((k & 4) ? sz : 0u);
                v[k] = *reinterpret_cast<const f4u*>(ls.src + lin * ls.stride + ls.off);
              }
              asm volatile("" ::: "memory");  // (keeps the two branches' loads apart: merged, they lose the scalar-base addressing)
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {  // two elements per packed fused multiply-add (v_pk_fma_f32)
              const vf2 lo = {v[k].v[0], v[k].v[1]}, hi = {v[k].v[2], v[k].v[3]}, wk = {w[k], w[k]};
              a01 = __builtin_elementwise_fma(lo, wk, a01);
              a23 = __builtin_elementwise_fma(hi, wk, a23);
            }
          }
          prod[0] = ls.yb[0] * a01.x;
          prod[1] = ls.yb[1] * a01.y;
          prod[2] = ls.yb[2] * a23.x;
          prod[3] = ls.yb[3] * a23.y;
          if (g.layout != RF_LAYOUT_SPLIT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
              for (int ch = 0; ch < 3; ++ch) rgb[ch] += (ls.chan[j] == ch) ? prod[j] : 0.0f;
            }
          }
        }
      }
      if (!L::kCorner && g.layout == RF_LAYOUT_SPLIT) {  // (wave-uniform)
        if constexpr (!L::kCorner) {
          const float rest = rest_colour_sums<K>(prod);
          if (has && ls.closes_colour) my_rgb[slot * 4 + ls.colour] = rest;
        }
      } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) rgb[ch] = group_sum_to_last<LPS>(rgb[ch]);
        if (has && sub == LPS - 1) {
          if constexpr (L::kCorner) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) rgb[ch] = kC0 * rgb[ch];
          }
          my_rgb[slot * 4 + 0] = rgb[0];
          my_rgb[slot * 4 + 1] = rgb[1];
          my_rgb[slot * 4 + 2] = rgb[2];
        }
      }
    }
    wave_lds_fence();

    // ---------------- P2: lanes = samples ----------------
    // (lanes without `need` read some other sample's slot or stale words: never used)
    const float raw_r = fast_base ? fast_rgb[0] : (base_in_p0 ? fast_rgb[0] + my_rgb[slot * 4 + 0] : my_rgb[slot * 4 + 0]);
    const float raw_g = fast_base ? fast_rgb[1] : (base_in_p0 ? fast_rgb[1] + my_rgb[slot * 4 + 1] : my_rgb[slot * 4 + 1]);
    const float raw_b = fast_base ? fast_rgb[2] : (base_in_p0 ? fast_rgb[2] + my_rgb[slot * 4 + 2] : my_rgb[slot * 4 + 2]);
    if (need) {
      part_c[0] += w * sigmoidf_(raw_r);
      part_c[1] += w * sigmoidf_(raw_g);
      part_c[2] += w * sigmoidf_(raw_b);
    }
    if (sm.valid) {
      part_acc += w;
      part_depth += w * sm.z;
    }
    if constexpr (SAVE) {
      // `need` samples are exactly those that can carry gradient (inside, T != 0, and sigma != 0 under ReLU): only they are
      // cached -- compacted to the front of the chunk's 64 slots, so that the adjoint reads whole lines of useful entries --
      // flagged in the chunk's mask, and counted per key for the binned adjoint
      if (need) {
        const long long idx = ray * (long long)r.S + chunk * kWave + slot;
        float4 cv;
        cv.x = raw_r;
        cv.y = raw_g;
        cv.z = raw_b;
        cv.w = sigma;
        store_f4<RF_NT_CACHE_STORE>(reinterpret_cast<float4*>(out.cache) + idx, cv);
        if (RF_NT_CACHE_STORE)
          __builtin_nontemporal_store(T, out.tcache + idx);
        else
          out.tcache[idx] = T;
      }
      my_cmask = (lane == (chunk & (kWave - 1))) ? mask : my_cmask;
      if (out.hist) add_key_runs<false>(out.hist, need ? brick_key(sm.cell.i0, g, out.brick_shift, out.nby, out.nbz) : -1, lane);
    }
    wave_lds_fence();
    if (T_carry == 0.0f) break;  // every later weight is exactly 0
  }

  const float cr = wave_sum(part_c[0]), cg = wave_sum(part_c[1]), cb = wave_sum(part_c[2]);
  const float acc = wave_sum(part_acc), depth = wave_sum(part_depth);
  if (lane == 0) {
    const float bg = white ? (1.0f - acc) : 0.0f;
    out.colour[ray * 3 + 0] = white ? cr + bg : cr;
    out.colour[ray * 3 + 1] = white ? cg + bg : cg;
    out.colour[ray * 3 + 2] = white ? cb + bg : cb;
    out.depth[ray] = depth;
    out.acc[ray] = acc;
    const float q = depth / acc;  // 0/0 = NaN propagates like torch.maximum (accumulate.py:85-88)
    out.disparity[ray] = 1.0f / ((q != q) ? q : fmaxf(kZeroPlus, q));
    if constexpr (SAVE) out.stop[ray] = processed;
  }
  if constexpr (SAVE) {
    flush_cmasks(cmask_group);
    for (int gq = cmask_group + 1; gq * kWave < nchunks; ++gq) flush_cmasks(gq);  // (chunks never reached: zero masks)
  }
}

template <int K, bool DIFFUSE, bool SAVE>
__global__ __launch_bounds__(kBlock, RF_FWD_WAVES) void render_forward_kernel(GridArgs g, RayArgs r, OutArgs out, uint32_t flags) {
  __shared__ __attribute__((aligned(16))) uint32_t s_entry[kWavesPerBlock][kWave * kEntryFwd];
  __shared__ __attribute__((aligned(16))) float s_rgb[kWavesPerBlock][kWave * 4];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  render_forward_ray<K, DIFFUSE, SAVE>(g, r, out, flags, (long long)blockIdx.x * kWavesPerBlock + wave, lane, s_entry[wave], s_rgb[wave]);
}

// The two renders of a training iteration -- specular, and render_diffuse with its own jitter stream -- of the SAME rays in one launch:
// a block takes two rays, waves 0, 1 walk them for the specular render, waves 2, 3 for the diffuse one.  Sample s of both renders lies
// in the same stratum of the ray, i.e. in the same or a neighbouring cell: the second render finds the 16-byte base records of its
// corners in the CU's L1 / the XCD's L2 instead of fetching them from HBM a second time.
struct ForwardPair {
  RayArgs r[2];
  OutArgs out[2];
  uint32_t flags[2];
};
template <int K>
__global__ __launch_bounds__(kBlock, RF_FWD_WAVES) void render_forward_pair_kernel(GridArgs g, ForwardPair p) {
  __shared__ __attribute__((aligned(16))) uint32_t s_entry[kWavesPerBlock][kWave * kEntryFwd];
  __shared__ __attribute__((aligned(16))) float s_rgb[kWavesPerBlock][kWave * 4];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long ray = (long long)blockIdx.x * (kWavesPerBlock / 2) + (wave & 1);
  if (wave < 2) {
    if constexpr (K == 1)
      render_forward_ray<1, true, true>(g, p.r[0], p.out[0], p.flags[0], ray, lane, s_entry[wave], s_rgb[wave]);
    else
      render_forward_ray<K, false, true>(g, p.r[0], p.out[0], p.flags[0], ray, lane, s_entry[wave], s_rgb[wave]);
  } else {
    render_forward_ray<1, true, true>(g, p.r[1], p.out[1], p.flags[1], ray, lane, s_entry[wave], s_rgb[wave]);
  }
}

// =============================================================================================
// frame render, ray PACKETS: one wavefront = the 8 x 8 pixel tile of a posed-camera frame, lanes = rays, all of them at the SAME
// sample index s (VolumetricModel.render, modules/volumetric_model.py:143-172; inference only).
//
// The per-ray kernel above walks one ray per wave with lanes = samples: every lane gathers its own 8 corners (8 L1 look-ups per
// sample for the base records, 48 more per sample that needs its features), and on a coherent frame it is bound by vector-ALU issue
// (81 % busy) with the L1's line look-up rate as the co-limit -- although the 64 rays of a pixel tile at one
// sample index sit within ~2.5 voxels of each other and ask for the same few nodes.  Here the wave fetches the NEIGHBOURHOOD once:
// per step the 64 lanes load one node each of a 4 x 4 x 4-node window (16-byte base record + 96-byte rest record, 7 loads per lane
// whatever the lanes need) into LDS, and every lane interpolates its own sample from LDS (8 + 48 ds_read_b128).  Compositing is the
// lane's own sequential recurrence T <- T (1 - alpha) -- torch.cumprod's order (accumulate.py:66-67) -- with no cross-lane scan.
// The window: origin = per-axis minimum of the lanes' lower nodes, pulled towards an anchor lane so that the anchor is always
// covered; lanes whose cell does not fit take part in another round with a window of their own (a tile that straddles more
// than three cells on an axis: fine grids, grazing views, per-ray AABB bounds).  Per-sample arithmetic -- z, point, cell, weights,
// the density's separately rounded products in ATen's corner order, the fused colour sums in the per-ray kernel's order, alpha,
// sigmoid -- is the per-ray kernel's, value for value; only the order in which a ray's weighted samples are ADDED differs
// (sequential here, per-lane partial sums + a butterfly there), so the two kernels agree to summation order, not bit for bit.
// A pixel's result does not depend on the other pixels of its tile or on how a frame is cut into calls.
// Split / bricked storage, every SH degree (degree 2 = the tuned path; the base record alone for degree 0 and render_diffuse).
// =============================================================================================
#ifndef RF_TILE_WAVES
#define RF_TILE_WAVES 4
#endif
// LDS layout of the window: node (x, y, z) of the 4 x 4 x 4 window sits in slot x * kTileXS + y * 4 + z.  A ds_read_b128 covers four of
// the 64 banks: with 16-byte base records two slots collide when they differ by a multiple of 16, with the 96-byte rest records by a
// multiple of 8.  The plain stride 16 makes every pair of nodes one step apart in x collide (41 % of the LDS pipe's active cycles
// were bank conflicts: profiles/r06_frame_packets_counters.md); with stride 18 no two DIFFERENT nodes of a unit neighbourhood
// (offsets in {-1, 0, 1}^3) collide, in either record size: 18 dx + 4 dy + dz is never 0 mod 8 for them.
constexpr int kTileXS = 18;
constexpr int kTileSlots = (3 * kTileXS + 16 + 3) / 4 * 4;  // (whole multiples of four slots)
// slot of the node lane 16 x + 4 y + z loads: lane + x (kTileXS - 16).  (Written out at its three uses: as a function the same expression
// sends the K = 9 instantiation from 2 to 34 spilled registers -- the kernel sits on its 128-register budget.)
#define RF_TILE_SLOT(lane) ((lane) + ((lane) >> 4) * (kTileXS - 16))
// a lane's cell in (clamped) node coordinates -- corners_of's rules: lower node c0, step e to the upper node (0 where clamping
// collapses the two), the 8 trilinear weights with out-of-grid nodes forced to 0
struct TileCell {
  int c0[3], e[3];
  float w[8];
};
__device__ __forceinline__ TileCell tile_cell(const Cell& c, const GridArgs& g) {
  const int dims3[3] = {g.X, g.Y, g.Z};
  TileCell t;
  float aw[3][2];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int i0 = c.i0[a];
    const bool ok0 = i0 >= 0, ok1 = i0 + 1 < dims3[a];
    t.c0[a] = max(i0, 0);
    t.e[a] = (ok0 && ok1) ? 1 : 0;
    aw[a][0] = ok0 ? c.w0[a] : 0.0f;
    aw[a][1] = ok1 ? c.w1[a] : 0.0f;
  }
  const float wxy[4] = {aw[0][0] * aw[1][0], aw[0][1] * aw[1][0], aw[0][0] * aw[1][1], aw[0][1] * aw[1][1]};  // [dx + 2 dy]
#pragma unroll
  for (int k = 0; k < 8; ++k) t.w[k] = wxy[k & 3] * aw[2][k >> 2];
  return t;
}

// origin of the 4 x 4 x 4-node window for the lanes that still wait: the per-axis minimum of their lower nodes, pulled up towards an
// anchor lane (the tile's centre ray while it waits, else the first waiting one) so that the anchor's cell always fits
// (= max(min over the waiting lanes, anchor - 2) per axis.  The anchor waits itself, so the minimum is at most the anchor's node and the
// result is one of anchor - 2, anchor - 1, anchor: two wave votes per axis -- "does a waiting lane lie below the anchor", "... below
// anchor - 1" -- instead of a six-step cross-lane minimum.)
__device__ __forceinline__ void tile_window_origin(bool pending, unsigned long long pm, const int c0[3], int O[3]) {
  const int anchor = ((pm >> 27) & 1ull) ? 27 : __builtin_ctzll(pm);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int A = __builtin_amdgcn_readlane(c0[a], anchor);
    const bool below1 = __ballot(pending && c0[a] < A) != 0ull, below2 = __ballot(pending && c0[a] < A - 1) != 0ull;
    O[a] = A - (below1 ? 1 : 0) - (below2 ? 1 : 0);
  }
}

// the window in registers, one node per lane (node O + (lane >> 4, (lane >> 2) & 3, lane & 3), clamped into the grid).  (Named
// members filled in place, no arrays, no struct copies: the compiler keeps an aggregate that is assigned under a condition in scratch
// memory.)
struct TileWindow {
  vf4 b, q0, q1, q2, q3, q4, q5;  // (native vectors: HIP's float4 is a class with a union inside, which kept these in scratch memory)
};
__device__ __forceinline__ unsigned int tile_window_node(const GridArgs& g, const int O[3], int lane) {
  const int nx = min(O[0] + (lane >> 4), g.X - 1), ny = min(O[1] + ((lane >> 2) & 3), g.Y - 1), nz = min(O[2] + (lane & 3), g.Z - 1);
  return node_lin(g, nx, ny, nz);
}
// (the per-ray kernels' "near" addressing where the host allows it -- scalar 64-bit base + a 32-bit byte offset built with a full-rate
// 24-bit multiply --, else general 64-bit addresses: a 64-bit multiply-add per record issues at a fraction of the rate and the kernel
// is bound by vector-ALU issue)
__device__ __forceinline__ const char* tile_record(const float* tensor, long long stride, unsigned int lin, int near32) {
  if (near32) return reinterpret_cast<const char*>(tensor) + (size_t)__umul24(lin, (unsigned int)stride * 4u);
  return reinterpret_cast<const char*>(tensor + (size_t)lin * (size_t)stride);
}
__device__ __forceinline__ void tile_window_load_base(TileWindow& wdw, const GridArgs& g, unsigned int lin) {
  wdw.b = *reinterpret_cast<const vf4*>(tile_record(g.dens, g.dstride, lin, g.near32));
}
__device__ __forceinline__ void tile_window_load_rest(TileWindow& wdw, const GridArgs& g, unsigned int lin) {
  const char* rp = tile_record(g.feat, g.fstride, lin, g.near32);
  wdw.q0 = *reinterpret_cast<const vf4*>(rp);
  wdw.q1 = *reinterpret_cast<const vf4*>(rp + 16);
  wdw.q2 = *reinterpret_cast<const vf4*>(rp + 32);
  wdw.q3 = *reinterpret_cast<const vf4*>(rp + 48);
  wdw.q4 = *reinterpret_cast<const vf4*>(rp + 64);
  wdw.q5 = *reinterpret_cast<const vf4*>(rp + 80);
}
__device__ __forceinline__ void tile_window_store_rest(const TileWindow& wdw, vf4* my_rest, int lane) {
  vf4* d = my_rest + RF_TILE_SLOT(lane) * 6;
  d[0] = wdw.q0;
  d[1] = wdw.q1;
  d[2] = wdw.q2;
  d[3] = wdw.q3;
  d[4] = wdw.q4;
  d[5] = wdw.q5;
}

// the lanes whose cell lies inside the window interpolate their sample from LDS -- first the base record (density, degree-0 colour);
// returns whether this lane was one of them, and in `nk` the window slots of its 8 corners
__device__ __forceinline__ bool tile_interpolate_base(bool pending, const TileCell& tc, const int O[3], const vf4* my_base, const GridArgs& g, float T, int nk[8],
                                                      float& sigma, float raw[3], bool& need) {
  const int l0[3] = {tc.c0[0] - O[0], tc.c0[1] - O[1], tc.c0[2] - O[2]};
  // (O is the minimum over the waiting lanes unless the anchor pulled it up: then the lanes below it wait for another round)
  const bool covered = pending && l0[0] >= 0 && l0[1] >= 0 && l0[2] >= 0 && l0[0] + tc.e[0] <= 3 && l0[1] + tc.e[1] <= 3 && l0[2] + tc.e[2] <= 3;
  const int n0 = l0[0] * kTileXS + l0[1] * 4 + l0[2];
  const int ex = tc.e[0] * kTileXS, ey = tc.e[1] * 4, ez = tc.e[2];
#pragma unroll
  for (int k = 0; k < 8; ++k) nk[k] = n0 + ((k & 1) ? ex : 0) + ((k & 2) ? ey : 0) + ((k & 4) ? ez : 0);
  if (covered) {
    // density in the reference's operation order (separately rounded products, ATen's corner order); degree-0 colour fused
    float acc = 0.0f, cb = 0.0f;
    vf2 crg = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const vf4 t = my_base[nk[k]];
      float v = t[0] * g.rho;
      if (g.mode == RF_DENSITY_ABS) v = fabsf(v);
      acc = acc + v * tc.w[k];
      const vf2 rg = {t[1], t[2]}, wk = {tc.w[k], tc.w[k]};
      crg = __builtin_elementwise_fma(rg, wk, crg);
      cb = __builtin_fmaf(t[3], tc.w[k], cb);
    }
    sigma = density_post(acc, g.mode);
    raw[0] = kC0 * crg.x;
    raw[1] = kC0 * crg.y;
    raw[2] = kC0 * cb;
    need = (T != 0.0f) && !(g.mode == RF_DENSITY_RELU && sigma == 0.0f);
  }
  return covered;
}
// ... then, for the lanes that need their colour, the 24 rest coefficients: element c * 8 + (k - 1), each summed over the corners with
// fused multiply-adds in corner order, then basis * sum and the per-ray kernel's grouping ((p1 + p2) + (p3 + p4)) + ((p5 + p6) + (p7 + p8))
__device__ __forceinline__ void tile_interpolate_rest(const TileCell& tc, const int nk[8], const vf4* my_rest, const float Y[16], float raw[3]) {
  constexpr int KR = 8;
  vf2 a2[12];
#pragma unroll
  for (int t = 0; t < 12; ++t) a2[t] = vf2{0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const vf2 wk = {tc.w[k], tc.w[k]};
    const vf4* rec = my_rest + __umul24((unsigned)nk[k], 6u);  // (a full-rate 24-bit multiply: the plain 32-bit one issues at quarter rate)
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const vf4 v = rec[t];
      a2[2 * t] = __builtin_elementwise_fma(vf2{v[0], v[1]}, wk, a2[2 * t]);
      a2[2 * t + 1] = __builtin_elementwise_fma(vf2{v[2], v[3]}, wk, a2[2 * t + 1]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float p[KR];
#pragma unroll
    for (int jj = 0; jj < KR; ++jj) {
      const int el = c * KR + jj;
      const float av = (el & 1) ? a2[el >> 1].y : a2[el >> 1].x;
      p[jj] = Y[1 + jj] * av;
    }
    const float d0 = (p[0] + p[1]) + (p[2] + p[3]), d1 = (p[4] + p[5]) + (p[6] + p[7]);
    raw[c] = raw[c] + (d1 + d0);
  }
}

// SH degree 1 / 3 (K = 4 / 16): rest records of 3 (K - 1) = 9 / 45 floats -- not whole quads, 4-byte aligned.  Always fetched on
// demand (after the densities said that a lane needs colours): whole quads by 16-byte loads at 4-byte alignment, the odd float by
// itself (nothing is read behind a record: the last node's record ends the tensor), stored as kQ = 3 / 12 quads per window slot.
template <int K>
struct TileRestGeneric {
  static constexpr int kKR = K - 1, kF = 3 * kKR, kFull = kF / 4, kRem = kF % 4, kQ = (kF + 3) / 4;
};
template <int K>
__device__ __forceinline__ void tile_window_rest_generic(const GridArgs& g, unsigned int lin, vf4* my_rest, int lane) {
  using R = TileRestGeneric<K>;
  const float* rp = reinterpret_cast<const float*>(tile_record(g.feat, g.fstride, lin, g.near32));
  f4u q[R::kFull];
  float tail[R::kRem > 0 ? R::kRem : 1];
#pragma unroll
  for (int t = 0; t < R::kFull; ++t) q[t] = *reinterpret_cast<const f4u*>(rp + 4 * t);
#pragma unroll
  for (int t = 0; t < R::kRem; ++t) tail[t] = rp[4 * R::kFull + t];
  vf4* d = my_rest + RF_TILE_SLOT(lane) * R::kQ;
#pragma unroll
  for (int t = 0; t < R::kFull; ++t) d[t] = vf4{q[t].v[0], q[t].v[1], q[t].v[2], q[t].v[3]};
  if constexpr (R::kRem > 0) d[R::kFull] = vf4{tail[0], R::kRem > 1 ? tail[R::kRem > 1 ? 1 : 0] : 0.0f, R::kRem > 2 ? tail[R::kRem > 2 ? 2 : 0] : 0.0f, 0.0f};
}
// element c * (K - 1) + (k - 1) of the record, summed over the corners with fused multiply-adds in corner order, then basis * sum,
// added up pairwise
template <int K>
__device__ __forceinline__ void tile_interpolate_rest_generic(const TileCell& tc, const int nk[8], const vf4* my_rest, const float Y[16], float raw[3]) {
  using R = TileRestGeneric<K>;
  vf2 a2[2 * R::kQ];
#pragma unroll
  for (int t = 0; t < 2 * R::kQ; ++t) a2[t] = vf2{0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const vf2 wk = {tc.w[k], tc.w[k]};
    const vf4* rec = my_rest + __umul24((unsigned)nk[k], (unsigned)R::kQ);
#pragma unroll
    for (int t = 0; t < R::kQ; ++t) {
      const vf4 v = rec[t];
      a2[2 * t] = __builtin_elementwise_fma(vf2{v[0], v[1]}, wk, a2[2 * t]);
      a2[2 * t + 1] = __builtin_elementwise_fma(vf2{v[2], v[3]}, wk, a2[2 * t + 1]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float p[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int el = c * R::kKR + (jj < R::kKR ? jj : 0);
      const float av = (el & 1) ? a2[el >> 1].y : a2[el >> 1].x;
      p[jj] = jj < R::kKR ? Y[1 + (jj < R::kKR ? jj : 0)] * av : 0.0f;
    }
#pragma unroll
    for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
      for (int jj = 0; jj < w; ++jj) p[jj] = p[jj] + p[jj + w];
    raw[c] = raw[c] + p[0];
  }
}

// this lane's ray of the tile: pixel (i, j), cast_rays fused (utils/misc.py:12-50) -- lanes off the frame or outside the pixel
// range of the call compute on a clamped pixel and never write
__device__ __forceinline__ RayState tile_ray(const GridArgs& g, const RayArgs& r, uint32_t flags, int i, int j) {
  RayState st;
  const float R[9] = {r.pose[0], r.pose[1], r.pose[2], r.pose[4], r.pose[5], r.pose[6], r.pose[8], r.pose[9], r.pose[10]};
  pixel_ray(i, j, r.H, r.W, r.focal, R, st.d);
  st.o[0] = r.pose[3];
  st.o[1] = r.pose[7];
  st.o[2] = r.pose[11];
  st.jseed = r.jitter ? jitter_ray_seed(r.jkey, (long long)i * r.W + j) : 0u;
  st.dnorm = sqrtf((st.d[0] * st.d[0] + st.d[1] * st.d[1]) + st.d[2] * st.d[2]);
  st.near = r.near;
  st.far = r.far;
  if (flags & RF_FLAG_AABB_SAMPLING) {
    float t0, t1;
    ray_box(st.o, st.d, g.amin, g.amax, r.near, r.far, t0, t1);
    st.near = t0;
    st.far = t1;
  }
  return st;
}

// z_of for a frame: the keyed jitter or none (frames with a caller-supplied jitter TABLE are the per-ray kernel's: rf_render_forward)
__device__ __forceinline__ float tile_z(const RayState& st, const RayArgs& r, int s) {
  const int sc = min(s, r.S - 1);
  const float u = r.jitter ? jitter_uniform(st.jseed, sc) : 0.0f;
  return z_sample(r.tvals, r.jitter != 0, u, sc, r.S, st.near, st.far);
}

// (A variant pipelined over the sample index -- the window of sample s + 1 requested into registers before sample s is interpolated, two
// waves per SIMD with up to 256 registers each -- measured SLOWER: 2.54 against 2.01 ms per 800 x 800 x 256 frame; at four waves per SIMD it
// spills 176 registers.  Four resident waves hide the window's round trip better than one wave's own prefetch.)
// Tile of a wave.  WPB waves per workgroup (nothing is shared between them: WPB = 1 lets the dispatcher place every tile by itself);
// XCD_ROWS: workgroup b runs on XCD b % 8 (observed placement, used for speed only: MI355X_MICROARCH.md, workgroup dispatch) -- the
// map gives each XCD whole tile rows, interleaved (row = 8 * group + b % 8): a tile's neighbours along its row go through the same
// L2 while the eight XCDs still share every part of the picture (the heavy tiles sit in its middle).
// Measured (profiles/r06_frame_tile_sched.txt), 800 x 800: WPB 1 + XCD rows against WPB 4 + linear: 128^3 / 256 samples 1.69-1.73
// against 1.70-1.75 ms, 256^3 / 512 samples with the mask 1.52-1.55 against 1.62-1.66 ms; either switch alone: no gain.  The gain is
// placement granularity (a workgroup of four tiles holds its LDS and slot until its slowest tile is done), not cache locality: the
// counters show MORE L2 requests with one wave per workgroup (the four tiles no longer share a CU's L1) at the same kernel time --
// memory is not what this kernel waits for (profiles/r06_frame_packets_counters.md).
// K = the coefficients per colour that are read: 1 (SH degree 0, render_diffuse: the base record alone), 4, 9, 16.  K = 9 is the tuned
// path (rest records = six aligned quads, prefetched with the base records where the previous step needed colours); K = 4 / 16 use the
// generic rest path above (K = 16: 46 accumulator registers per lane -- two waves per SIMD instead of four).
template <int K, int WPB, bool XCD_ROWS>
__global__ __launch_bounds__(kWave * WPB, (K == 16 ? 2 : RF_TILE_WAVES)) void render_frame_tile_kernel(GridArgs g, RayArgs r, OutArgs out, uint32_t flags, int row0, int tile_rows, int tiles_x) {
  constexpr bool REST = K > 1;
  constexpr int kRestQuads = K == 9 ? 6 : (K > 1 ? TileRestGeneric<K>::kQ : 0);
  __shared__ __attribute__((aligned(16))) vf4 s_base[WPB][kTileSlots];
  __shared__ __attribute__((aligned(16))) vf4 s_rest[WPB][REST ? kTileSlots * kRestQuads : 1];
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int ty, tx;
  if constexpr (XCD_ROWS) {
    const int segs_x = (tiles_x + WPB - 1) / WPB;  // workgroups per tile row
    const int b = (int)blockIdx.x, group = b / (8 * segs_x), within = b - group * (8 * segs_x);
    ty = group * 8 + (within & 7);
    tx = (within >> 3) * WPB + wave;
    if (tx >= tiles_x) return;
  } else {
    const int tile = (int)blockIdx.x * WPB + wave;
    ty = tile / tiles_x;
    tx = tile - ty * tiles_x;
  }
  if (ty >= tile_rows) return;
  vf4* my_base = s_base[wave];
  vf4* my_rest = s_rest[wave];

  // lane -> pixel of the 8 x 8 tile.  ds_read_b128 is serviced in four fixed groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19,
  // 28-31} and the same + 32: MI355X_MICROARCH.md, LDS): each group renders one 4 x 4 QUADRANT of the tile, so the lanes that read the
  // window in the same LDS cycle ask for the fewest distinct nodes (two nodes one step apart in x share banks).  A pixel's result
  // does not depend on its lane.
  const int l5 = lane & 31;
  const bool second = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;             // which of the half's two groups
  const int in_group = second ? (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16)) : (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12));
  const int quadrant = (lane >> 5) * 2 + (second ? 1 : 0);
  const int i_raw = row0 + ty * 8 + (quadrant >> 1) * 4 + (in_group >> 2), j_raw = tx * 8 + (quadrant & 1) * 4 + (in_group & 3);
  const int i = min(i_raw, r.H - 1), j = min(j_raw, r.W - 1);
  const long long ray_raw = ((long long)i * r.W + j) - r.ray0;
  const bool lane_valid = i_raw < r.H && j_raw < r.W && ray_raw >= 0 && ray_raw < r.n;
  // (the output index; the <true> instantiations sit at their 128-register budget and park these two registers in scratch memory
  // ACROSS the march -- one store before it, one load behind it per wave, nothing inside; every attempt to rebuild the index after
  // the march instead moved a spill INTO the loop)
  const long long ray = ray_raw;
  const RayState st = tile_ray(g, r, flags, i, j);
  const bool white = flags & RF_FLAG_WHITE_BKGD;
  const bool use_occ = (flags & RF_FLAG_OCCUPANCY_SKIP) && g.occ != nullptr;
  float Y[16];
  if constexpr (REST) sh_basis<K>(st.d[0] / st.dnorm, st.d[1] / st.dnorm, st.d[2] / st.dnorm, Y);  // v = d / |d| (process.py:53)
  // the ray's parameter interval inside the box, widened by the per-ray kernel's margins (slab interval, one stratum of jitter) -- two
  // registers for the march: [z_lo, z_hi], empty for lanes that never write and rays that miss the box
  float z_lo = kInfinity, z_hi = -kInfinity;
  {
    const BoxSpan span = box_span(st, r, g);
    if (lane_valid && span.hits) {
      z_lo = (span.t_in - span.margin) - span.zpad;
      z_hi = (span.t_out + span.margin) + span.zpad;
    }
  }

  float T = 1.0f;
  float part_c[3] = {0.f, 0.f, 0.f};
  float part_acc = 0.f, part_depth = 0.f;
  float z_cur = 0.0f;
  bool z_ready = false;
  bool rest_hot = true;  // (wave-uniform) did the previous step with live lanes need colours?

  for (int s = 0; s < r.S; ++s) {
    // ---- can any ray of the tile be inside the box at this sample index?  (a sample outside contributes exactly nothing --
    // sigma = 0, alpha = 0, w = 0, T unchanged)
    {
      const float zc = z_uniform(st.near, st.far, r.tvals[s]);
      const bool maybe = zc >= z_lo && zc <= z_hi;
      if (__ballot(maybe) == 0ull) {
        z_ready = false;
        continue;
      }
    }
    if (!z_ready) z_cur = tile_z(st, r, s);
    const float z_next = tile_z(st, r, s + 1);
    const Sample sm = sample_at(st, r, g, s, z_cur, z_next);
    z_cur = z_next;
    z_ready = true;
    bool live = lane_valid && sm.inside;
    if (use_occ && live) live = cell_occupied(sm.cell, g);
    float sigma = 0.0f;
    float raw[3] = {0.f, 0.f, 0.f};
    bool need = false;
    bool pending = live;
    unsigned long long pm = __ballot(pending);
    if (pm != 0ull) {
      const TileCell tc = tile_cell(sm.cell, g);
      bool step_needs_rest = false;
      do {
        int O[3];
        tile_window_origin(pending, pm, tc.c0, O);
        // the 4 x 4 x 4-node window, one node per lane: global -> registers -> LDS.  The 96-byte rest records travel with the base
        // records when the previous step of this tile needed colours (`rest_hot`: they are needed in contiguous regions), else
        // only once the interpolated densities say that a lane of this round does -- empty space costs one 16-byte load per lane
        const unsigned int lin = tile_window_node(g, O, lane);
        TileWindow wdw;
        tile_window_load_base(wdw, g, lin);
        const bool with_rest = K == 9 && rest_hot;  // (wave-uniform)
        if constexpr (K == 9) {
          if (with_rest) tile_window_load_rest(wdw, g, lin);
        }
        my_base[RF_TILE_SLOT(lane)] = wdw.b;
        if constexpr (K == 9) {
          if (with_rest) tile_window_store_rest(wdw, my_rest, lane);
        }
        wave_lds_fence();
        int nk[8];
        const bool covered = tile_interpolate_base(pending, tc, O, my_base, g, T, nk, sigma, raw, need);
        if constexpr (REST) {
          const bool mine = covered && need;
          if (__ballot(mine) != 0ull) {
            step_needs_rest = true;
            if constexpr (K == 9) {
              if (!with_rest) {
                tile_window_load_rest(wdw, g, lin);
                tile_window_store_rest(wdw, my_rest, lane);
                wave_lds_fence();
              }
              if (mine) tile_interpolate_rest(tc, nk, my_rest, Y, raw);
            } else {
              tile_window_rest_generic<K>(g, lin, my_rest, lane);
              wave_lds_fence();
              if (mine) tile_interpolate_rest_generic<K>(tc, nk, my_rest, Y, raw);
            }
          }
        }
        wave_lds_fence();  // (the next round overwrites the window)
        pending = pending && !covered;
        pm = __ballot(pending);
      } while (pm != 0ull);  // lanes whose cell did not fit take part in another round, with a window of their own
      rest_hot = step_needs_rest;
    }
    // ---- compositing, this ray's own recurrence (accumulate.py:63-88)
    float E;
    const float alpha = occupancy_alpha(sigma * sm.delta, E);
    const float wgt = alpha * T;
    if (need) {
      part_c[0] += wgt * sigmoidf_(raw[0]);
      part_c[1] += wgt * sigmoidf_(raw[1]);
      part_c[2] += wgt * sigmoidf_(raw[2]);
    }
    part_acc += wgt;
    part_depth += wgt * sm.z;
    T = T * E;
    if (__ballot(lane_valid && T != 0.0f) == 0ull) break;  // every later weight of every ray of the tile is exactly 0
  }
  if (lane_valid) {
    const float bg = white ? (1.0f - part_acc) : 0.0f;
    out.colour[ray * 3 + 0] = white ? part_c[0] + bg : part_c[0];
    out.colour[ray * 3 + 1] = white ? part_c[1] + bg : part_c[1];
    out.colour[ray * 3 + 2] = white ? part_c[2] + bg : part_c[2];
    out.depth[ray] = part_depth;
    out.acc[ray] = part_acc;
    const float q = part_depth / part_acc;  // 0/0 = NaN propagates like torch.maximum (accumulate.py:85-88)
    out.disparity[ray] = 1.0f / ((q != q) ? q : fmaxf(kZeroPlus, q));
  }
}

// The cached samples of a chunk (forward pass, SAVE): lane l of the adjoint handles sample 64 c + l like the forward pass did; its
// cache entry, if bit l of the chunk's mask is set, is entry rank(l) = popcount(mask below l) of the chunk's slots.  Lanes without
// an entry load entry 0 of the chunk (a line the wave touches anyway; a load under a lane condition would be followed by a
// register merge that waits for it) and are zeroed by the caller.
struct CachedSample {
  float4 cv;  // raw r, g, b, sigma
  float T;
};
__device__ __forceinline__ long long cached_slot(long long ray, int S, int chunk, unsigned long long cm, int lane) {
  const bool has = (cm >> lane) & 1ull;
  const int rank = __popcll(cm & ((1ull << lane) - 1ull));
  return ray * (long long)S + chunk * kWave + (has ? rank : 0);
}
// wave-uniform mask of chunk `chunk` out of the per-lane copies (lane c holds chunk 64 g + c)
__device__ __forceinline__ unsigned long long chunk_mask_of(unsigned long long lanes_masks, int chunk) {
  const int c = chunk & (kWave - 1);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)lanes_masks, c);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(lanes_masks >> 32), c);
  return ((unsigned long long)hi << 32) | lo;
}

// lanes per corner for the backward scatter: F features + 1 density, rounded up to a power of two
template <int K, bool DIFFUSE>
struct ScatterLayout {
  static constexpr bool kCorner = DIFFUSE || K == 1;  // only the degree-0 coefficient of each colour
  static constexpr int kF = 3 * K;
  static constexpr int kLPC = kCorner ? 4 : (kF + 1 <= 16 ? 16 : (kF + 1 <= 32 ? 32 : 64));
};

// =============================================================================================
// backward: dL/d(densities), dL/d(features) from dL/d(colour, depth, acc)
//
// With e_i = sum_ch gC[ch] (c_i[ch] - [white]) + gD z_i + gA :
//   dL/dR_i[ch] = w_i gC[ch] c_i (1 - c_i)
//   dL/dsigma_i = delta_i ( T_{i+1} e_i - sum_{j>i} w_j e_j ),   T_{i+1} = T_i (1 - alpha_i)   (division-free)
// Chunks are walked from the far end so that the suffix sum is an exact running sum.
// =============================================================================================
// EMIT: 0 = atomic scatter into the gradient tensors; 1 = per-slot keys + 32-byte records (binned backward, sort or
// scatter binning).  (Expanded records written straight to their final positions: render_emit_direct_kernel below.)
template <int K, bool DIFFUSE, int EMIT>
__global__ __launch_bounds__(kBlock) void render_backward_kernel(GridArgs g, RayArgs r, OutArgs fwd, GradArgs gr,
                                                                 uint32_t flags) {
  using SL = ScatterLayout<K, DIFFUSE>;

  __shared__ __attribute__((aligned(16))) uint32_t s_entry[kWavesPerBlock][kWave * kEntryBwd];

  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long ray = (long long)blockIdx.x * kWavesPerBlock + wave;
  if (ray >= r.n) return;

  const RayState st = load_ray(r, g, ray, flags);
  const float white = (flags & RF_FLAG_WHITE_BKGD) ? 1.0f : 0.0f;
  float gC[3] = {0.f, 0.f, 0.f};
  if (gr.gcolour) {
    gC[0] = gr.gcolour[ray * 3 + 0];
    gC[1] = gr.gcolour[ray * 3 + 1];
    gC[2] = gr.gcolour[ray * 3 + 2];
  }
  const float gD = gr.gdepth ? gr.gdepth[ray] : 0.0f;
  const float gA = gr.gacc ? gr.gacc[ray] : 0.0f;
  const bool no_upstream = gC[0] == 0.f && gC[1] == 0.f && gC[2] == 0.f && gD == 0.f && gA == 0.f;
  if (EMIT == 1) {
    // every sample slot of the ray gets a key (the sort runs over the dense array)
    const int done = (no_upstream) ? 0 : fwd.stop[ray];
    for (int s2 = ((done + kWave - 1) / kWave) * kWave + lane; s2 < r.S; s2 += kWave) gr.keys[ray * (long long)r.S + s2] = kNoBrick;
  }
  if (no_upstream) return;

  // Scatter layout (measured on MI355X, tools/atomic_microbench.hip): float32 atomics retire at a fixed rate of
  // ~21 G 64-byte-sector requests/s however many dwords a request carries, so one instruction should cover as
  // many CONSECUTIVE dwords as possible.  Lane = channel: LPC lanes span one corner's F features + its density,
  // and the two z-neighbours (adjacent records in memory) always share an instruction.
  //   unit = (sample, corner);  corner q: bit0 = dz, bit1 = dy, bit2 = dx
  constexpr int LPC = SL::kLPC;          // lanes per corner
  constexpr int UPI = kWave / LPC;       // units per instruction
  constexpr int SPI = UPI >= 8 ? UPI / 8 : 1;  // samples per scatter iteration
  constexpr int NPASS = UPI >= 8 ? 1 : 8 / UPI;
  const int c = lane % LPC;   // channel slot of this lane
  const int unit = lane / LPC;
  // per-lane, per-ray constants: destination (pointer, voxel stride, offset in the record), which colour feeds this
  // channel (3 = density lane, -1 = idle) and its SH weight
  int lane_colour = -1;
  float lane_basis_w = 0.0f;
  float* lane_dst = gr.gfeat;
  long long lane_stride = g.fstride;
  int lane_off = 0;
  const bool split = g.layout == RF_LAYOUT_SPLIT;
  if constexpr (SL::kCorner) {
    const int kfull = g.F / 3;
    lane_basis_w = kC0;
    if (split) {  // base record: (sigma, r, g, b)
      lane_colour = (c == 0) ? 3 : c - 1;
      lane_dst = gr.gdens;
      lane_stride = g.dstride;
      lane_off = c;
    } else {
      lane_colour = c;  // 0,1,2 colour; 3 density
      lane_off = (c < 3) ? c * kfull : 0;
      if (c == 3) {
        lane_dst = gr.gdens;
        lane_stride = g.dstride;
      }
    }
  } else {
    float Y[16];
    sh_basis<K>(st.d[0] / st.dnorm, st.d[1] / st.dnorm, st.d[2] / st.dnorm, Y);
    float* ldsY = reinterpret_cast<float*>(s_entry[wave]);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < K; ++k) ldsY[k] = Y[k];
    }
    wave_lds_fence();
    if (split) {
      if (c < 4) {  // base record
        lane_colour = (c == 0) ? 3 : c - 1;
        lane_basis_w = ldsY[0];
        lane_dst = gr.gdens;
        lane_stride = g.dstride;
        lane_off = c;
      } else if (c <= SL::kF) {  // rest record, index = colour * (K-1) + (k-1)
        const int rr = c - 4;
        lane_colour = rr / (K - 1);
        lane_basis_w = ldsY[rr % (K - 1) + 1];
        lane_off = rr;
      }
    } else {
      if (c < SL::kF) {
        lane_colour = c / K;
        lane_basis_w = ldsY[c % K];
        lane_off = c;
      } else if (c == SL::kF) {
        lane_colour = 3;
        lane_dst = gr.gdens;
        lane_stride = g.dstride;
      }
    }
    wave_lds_fence();
  }
  uint32_t* my_entry = s_entry[wave];

  const int processed = fwd.stop[ray];
  const int nchunks = (processed + kWave - 1) / kWave;
  const int mask_words = (r.S + kWave - 1) / kWave;
  float suffix = 0.0f;  // sum of w_j e_j over all samples beyond the current chunk
  unsigned long long lane_masks = 0ull;
  int masks_group = -1;

  for (int chunk = nchunks - 1; chunk >= 0; --chunk) {
    const int s = chunk * kWave + lane;
    if ((chunk >> 6) != masks_group) {  // the masks of (up to) 64 chunks at a time, one per lane
      masks_group = chunk >> 6;
      const int c = masks_group * kWave + lane;
      lane_masks = fwd.cmask[ray * (long long)mask_words + min(c, mask_words - 1)];
    }
    const unsigned long long cm = chunk_mask_of(lane_masks, chunk);
    // a chunk without cached samples (outside the box, empty space, behind the point where T reached 0) contributes nothing
    if (cm == 0ull) {
      if constexpr (EMIT == 1) {  // every slot still gets its "no record" key
        if (s < r.S) gr.keys[ray * (long long)r.S + s] = kNoBrick;
      }
      continue;
    }
    Sample sm = make_sample(st, r, g, ray, s);
    const bool have = (cm >> lane) & 1ull;
    float raw[3] = {0.f, 0.f, 0.f};
    float sigma = 0.f, T = 0.f;
    {
      const long long idx = cached_slot(ray, r.S, chunk, cm, lane);
      const float4 cv = reinterpret_cast<const float4*>(fwd.cache)[idx];
      const float Tl = fwd.tcache[idx];
      if (have) {
        raw[0] = cv.x;
        raw[1] = cv.y;
        raw[2] = cv.z;
        sigma = cv.w;
        T = Tl;
      }
    }
    // (T_{i+1} e_i is the sample's OWN term of d L / d sigma_i = delta_i (T_i exp(-sigma_i delta_i) e_i - sum_{j>i} w_j e_j): with the
    // exponential itself, not 1 - alpha -- that difference cancels to a relative error of 6e-8 / exp(-x), which the 1e10-long interval
    // of a ray's last sample (accumulate.py:49-52) turns into per cents of its gradient when the sample lies inside the volume)
    float E;
    const float alpha = occupancy_alpha(sigma * sm.delta, E);
    const float w = alpha * T;
    const float Tn = T * E;
    float c[3], e = gD * sm.z + gA;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      c[ch] = sigmoidf_(raw[ch]);
      e += gC[ch] * (c[ch] - white);
    }
    const float we = have ? w * e : 0.0f;
    const float incl = wave_incl_rscan_add(we, lane);
    const float after = (incl - we) + suffix;  // sum over samples strictly behind this one
    suffix += __shfl(incl, 0, kWave);

    float g_sigma = sm.delta * (Tn * e - after);
    float g_pre;
    if (g.mode == RF_DENSITY_RELU)
      g_pre = (sigma > 0.0f) ? g_sigma : 0.0f;
    else if (g.mode == RF_DENSITY_SOFTPLUS)
      g_pre = g_sigma * softplus_slope(sigma);
    else
      g_pre = g_sigma;
    float g_raw[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) g_raw[ch] = (w * gC[ch]) * (c[ch] * (1.0f - c[ch]));

    const bool live = have && sm.inside;
    const bool need = live && (g_pre != 0.f || g_raw[0] != 0.f || g_raw[1] != 0.f || g_raw[2] != 0.f);
    if constexpr (EMIT == 1) {
      short key_of_lane = kNoBrick;
      if (sm.valid) {
        const long long slot = ray * (long long)r.S + s;
        short key = kNoBrick;
        if (need) {
          key = (short)brick_key(sm.cell.i0, g, gr.brick_shift, gr.nby, gr.nbz);
          constexpr int KE = SL::kCorner ? 1 : K;
          float4* rec = reinterpret_cast<float4*>(gr.records) + slot * record_quads(KE);
          if constexpr (KE == 1) {
            rec[0] = make_float4(sm.cell.idx[0], sm.cell.idx[1], sm.cell.idx[2], 0.0f);
            rec[1] = make_float4(g_pre * g.rho, g_raw[0] * kC0, g_raw[1] * kC0, g_raw[2] * kC0);
          } else {
            rec[0] = make_float4(sm.cell.idx[0], sm.cell.idx[1], sm.cell.idx[2], g_pre * g.rho);
            rec[1] = make_float4(g_raw[0], g_raw[1], g_raw[2], st.d[0] / st.dnorm);
            rec[2] = make_float4(st.d[1] / st.dnorm, st.d[2] / st.dnorm, 0.0f, 0.0f);
          }
        }
        gr.keys[slot] = key;
        key_of_lane = key;
      }
      if (gr.hist) add_key_runs<false>(gr.hist, (int)key_of_lane, lane);
      continue;
    }
    const unsigned long long mask = __ballot(need);
    const int count = __popcll(mask);
    if (need) {
      const int slot = __popcll(mask & ((1ull << lane) - 1ull));
      uint32_t* en = my_entry + slot * kEntryBwd;
      en[0] = pack_cell(sm.cell);
      en[1] = (uint32_t)lane;
      en[2] = __float_as_uint(sm.cell.w0[0]);
      en[3] = __float_as_uint(sm.cell.w1[0]);
      en[4] = __float_as_uint(sm.cell.w0[1]);
      en[5] = __float_as_uint(sm.cell.w1[1]);
      en[6] = __float_as_uint(sm.cell.w0[2]);
      en[7] = __float_as_uint(sm.cell.w1[2]);
      en[8] = __float_as_uint(g_pre);
      en[9] = __float_as_uint(g_raw[0]);
      en[10] = __float_as_uint(g_raw[1]);
      en[11] = __float_as_uint(g_raw[2]);
    }
    wave_lds_fence();

    // scatter: lane = channel
    for (int base = 0; base < count; base += SPI) {
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        const int u = pass * UPI + unit;     // unit index within this group of SPI samples
        const int slot = base + (u >> 3);
        const int q = u & 7;
        if (slot < count && lane_colour >= 0) {
          const uint32_t* en = my_entry + slot * kEntryBwd;
          const uint32_t pk = en[0];
          const int dx = (q >> 2) & 1, dy = (q >> 1) & 1, dz = q & 1;
          const int ix = (int)(pk & 0x7ffu) - 1 + dx, iy = (int)((pk >> 11) & 0x7ffu) - 1 + dy, iz = (int)(pk >> 22) - 1 + dz;
          const bool ok = ix >= 0 && ix < g.X && iy >= 0 && iy < g.Y && iz >= 0 && iz < g.Z;
          const float wx = __uint_as_float(en[2 + dx]), wy = __uint_as_float(en[4 + dy]), wz = __uint_as_float(en[6 + dz]);
          const float wc = (wx * wy) * wz;
          if (ok) {
            const long long lin = node_lin(g, ix, iy, iz);
            if (lane_colour == 3) {
              float gv = (wc * __uint_as_float(en[8])) * g.rho;
              if (g.mode == RF_DENSITY_ABS) {
                const float dv = g.dens[lin * g.dstride] * g.rho;
                gv = (dv > 0.f) ? gv : ((dv < 0.f) ? -gv : 0.0f);
              }
              if (gv != 0.0f) unsafeAtomicAdd(lane_dst + lin * lane_stride + lane_off, gv);
            } else {
              const float gsel = __uint_as_float(en[9 + lane_colour]);
              const float gv = wc * (gsel * lane_basis_w);
              if (gv != 0.0f) unsafeAtomicAdd(lane_dst + lin * lane_stride + lane_off, gv);
            }
          }
        }
      }
    }
    wave_lds_fence();
  }
}

// =============================================================================================
// direct emit (training step): the adjoint of a render as EXPANDED records at their final positions
//
// Every sample the forward pass counted (flag in the sign bit of its cached transmittance) writes index quad + dL/d(interpolated
// channel) for all channels (SH basis of the ray multiplied in) at the next free position of its (brick, flags) key class.
// The per-chunk dependency chain -- cache load (HBM) -> key -> returning cursor atomic (L2) -> record stores -- is what bounds this
// kernel, not arithmetic or bandwidth (0.3 of the HBM peak by counters), so a ray's chunks are processed in groups (RF_EMIT_G): all
// cache loads first, then all cursor atomics back to back (their results are not touched yet), then the gradients far-to-near
// with the running suffix sum.  Measured on the bench step (specular / diffuse render): 0.098 / 0.073 ms; by ablation the geometry
// and gradient arithmetic alone took 0.036 / 0.034 ms, the cache loads ~0.03, the cursor atomics ~0.025, the record stores
// ~0.035 / 0.017 on top -- the components added up rather than overlapped (one chunk at a time: 0.107 / 0.069 ms).  The reason
// showed in the ISA: the cache loads sat under `s < processed` and the t_vals reads of every sample under `s > 0` / `s < S - 1`,
// and the compiler follows a load under a lane condition with a register merge that waits for it -- a dozen exposed memory
// latencies per group.  With every load of a group unconditional (clamped indices) and issued before the first use:
// 0.090 / 0.046 ms.  ~2500 instructions per ray remain: issue-bound.
// Also tried: transposing the records through LDS so that a store instruction writes whole contiguous records instead of 64
// 16-byte pieces of 64 different lines -- slower (0.113 / 0.078 ms): the L2 merges the partial lines at no cost that matters here.
// Round 4, DENSE lanes: only a third of the lanes of a chunk pass hold a cached sample (67 records per ray over three or four chunks),
// so a variant walked a ray's cached samples as one dense list, 64 per pass (lane -> rank among the cached samples -> chunk from
// the running counts of the chunk masks -> position by an n-th-set-bit search): one or two passes per ray instead of three or four,
// same results -- and slower, 0.0877 -> 0.093 ms (one pass in flight) / 0.096-0.103 (two): the search is a dependent chain in front
// of the cache loads, and the kernel is bound by its chains (cache load -> key -> cursor atomic -> store), not by lane occupancy.
// =============================================================================================
template <int K, bool DIFFUSE>
__device__ __forceinline__ void render_emit_direct_ray(const GridArgs& g, const RayArgs& r, const OutArgs& fwd, const GradArgs& gr, uint32_t flags, long long ray,
                                                       int lane) {
  // the forward pass's counters have been turned into offsets: clear them for the next iteration (every thread of the grid
  // takes part, including the waves without a ray)
  if (gr.hist_clear)
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < gr.nkeys; k += (long long)gridDim.x * blockDim.x) gr.hist_clear[k] = 0;
  if (ray >= r.n) return;

  const RayState st = load_ray(r, g, ray, flags);
  const float white = (flags & RF_FLAG_WHITE_BKGD) ? 1.0f : 0.0f;
  float gC[3] = {0.f, 0.f, 0.f};
  if (gr.gcolour) {
    gC[0] = gr.gcolour[ray * 3 + 0];
    gC[1] = gr.gcolour[ray * 3 + 1];
    gC[2] = gr.gcolour[ray * 3 + 2];
  }
  const float gD = gr.gdepth ? gr.gdepth[ray] : 0.0f;
  const float gA = gr.gacc ? gr.gacc[ray] : 0.0f;
  constexpr int KE = DIFFUSE ? 1 : K;  // diffuse lists carry the base channels only
  constexpr int QE = record_quads(KE);
  // the unit viewing direction (process.py:53) travels in the record: the brick pass evaluates the SH basis from it
  const float vdir[3] = {st.d[0] / st.dnorm, st.d[1] / st.dnorm, st.d[2] / st.dnorm};

  const int mask_words = (r.S + kWave - 1) / kWave;
  const int processed = fwd.stop[ray];
  const int nchunks = (processed + kWave - 1) / kWave;
  float suffix = 0.0f;  // sum of w_j e_j over all samples beyond the current chunk
  constexpr int G = RF_EMIT_G;  // chunks in flight
  // the chunk masks, one per lane: requested before anything that depends on `processed` (the first group walked is almost always
  // the last one of the ray: rays of at most 4096 samples have a single group)
  int masks_group = (mask_words - 1) >> 6;
  unsigned long long lane_masks = fwd.cmask[ray * (long long)mask_words + min(masks_group * kWave + lane, mask_words - 1)];

  // z of the sample behind the last chunk walked (lane 63 of a chunk takes its "next sample" from lane 0 of the chunk behind it, which
  // the previous group of the far-to-near walk evaluated: one z per sample instead of two)
  float z_edge = z_of(st, r, ray, nchunks * kWave);
  for (int c0 = nchunks - 1; c0 >= 0; c0 -= G) {
    float4 cv[G];
    float Tc[G], zz[G], zn[G];
    unsigned long long cm[G];
    // -- A0: the masks of the cached samples of chunks c0, c0 - 1, ...: a chunk without any is skipped
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int chunk = c0 - u;
      cm[u] = 0ull;
      if (chunk >= 0) {  // wave-uniform
        if ((chunk >> 6) != masks_group) {
          masks_group = chunk >> 6;
          const int c = masks_group * kWave + lane;
          lane_masks = fwd.cmask[ray * (long long)mask_words + min(c, mask_words - 1)];
        }
        cm[u] = chunk_mask_of(lane_masks, chunk);
      }
    }
    // -- A1: every load of the group -- sample caches, t_vals (and the jitter table, if one is used) -- before the first use.
    // All unconditional, with clamped indices: a load under a condition is followed by a register merge that waits for it, which
    // had serialised the four chunks' cache loads and the t_vals reads of every sample into as many exposed memory latencies.
    ZRequests zq[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const int chunk = max(c0 - u, 0);
      const int s = chunk * kWave + lane;
      const long long idx = cached_slot(ray, r.S, chunk, cm[u], lane);
      cv[u] = load_f4<RF_NT_CACHE_LOAD>(reinterpret_cast<const float4*>(fwd.cache) + idx);
      Tc[u] = RF_NT_CACHE_LOAD ? __builtin_nontemporal_load(fwd.tcache + idx) : fwd.tcache[idx];
      zq[u] = z_requests(r, s);
    }
#pragma unroll
    for (int u = 0; u < G; ++u) zz[u] = z_from(st, r, zq[u], ray, max(c0 - u, 0) * kWave + lane);
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const float up = dpp_move<kDppWaveShl1, 0xf>(0.0f, zz[u]);  // lane i <- lane i + 1
      zn[u] = (lane == kWave - 1) ? z_edge : up;
      z_edge = read_lane(zz[u], 0);  // the chunk in front of this one ends at it
    }
    // -- A2: geometry, keys and the cursor atomics of the group, back to back
    float dl[G], ix[G][3];
    int base_[G], hl_[G];
    bool counted[G];
#pragma unroll
    for (int u = 0; u < G; ++u) {
      base_[u] = hl_[u] = 0;
      counted[u] = false;
      dl[u] = ix[u][0] = ix[u][1] = ix[u][2] = 0.0f;
      if (cm[u] == 0ull) continue;
      const int s = (c0 - u) * kWave + lane;
      const Sample sm = sample_at(st, r, g, s, zz[u], zn[u]);
      dl[u] = sm.delta;
      ix[u][0] = sm.cell.idx[0];
      ix[u][1] = sm.cell.idx[1];
      ix[u][2] = sm.cell.idx[2];
      counted[u] = (cm[u] >> lane) & 1ull;
      const int key = counted[u] ? brick_key(sm.cell.i0, g, gr.brick_shift, gr.nby, gr.nbz) : -1;
      // one atomic per RUN of equal keys (add_key_runs, split: the returned base is only combined in phase B)
      const bool active = key >= 0;
      const int prev = __shfl_up(key, 1);
      const bool head = active && (lane == 0 || prev != key);
      const unsigned long long heads = __ballot(head);
      const unsigned long long ends = __ballot(head || !active);
      const unsigned long long above = (lane == 63) ? 0ull : (ends & ~((2ull << lane) - 1ull));
      const int run = (above ? __builtin_ctzll(above) : 64) - lane;
      const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
      hl_[u] = below ? 63 - __builtin_clzll(below) : 0;
      if (head) base_[u] = atomicAdd(&gr.cursor[key], run);
    }
    // -- B: gradients, far to near
#pragma unroll
    for (int u = 0; u < G; ++u) {
      if (cm[u] == 0ull) continue;
      const bool have = counted[u];
      const float sigma = have ? cv[u].w : 0.0f;
      const float T = have ? Tc[u] : 0.0f;
      float E;  // (the own term uses the exponential itself: see render_backward_kernel)
      const float alpha = occupancy_alpha(sigma * dl[u], E);
      const float w = alpha * T;
      const float Tn = T * E;
      const float raw[3] = {cv[u].x, cv[u].y, cv[u].z};
      float c[3], e = gD * zz[u] + gA;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        c[ch] = sigmoidf_(raw[ch]);
        e += gC[ch] * (c[ch] - white);
      }
      const float we = have ? w * e : 0.0f;
      const float incl = wave_incl_rscan_add(we, lane);
      const float after = (incl - we) + suffix;  // sum over samples strictly behind this one
      suffix += __shfl(incl, 0, kWave);
      const float g_sigma = dl[u] * (Tn * e - after);
      float g_pre;
      if (g.mode == RF_DENSITY_RELU)
        g_pre = (sigma > 0.0f) ? g_sigma : 0.0f;
      else if (g.mode == RF_DENSITY_SOFTPLUS)
        g_pre = g_sigma * softplus_slope(sigma);
      else
        g_pre = g_sigma;
      const int pos = __shfl(base_[u], hl_[u]) + (lane - hl_[u]);
      if (have) {
        // (a counted sample is inside the box; its record is written even when its gradient happens to vanish)
        const float graw[3] = {(w * gC[0]) * (c[0] * (1.0f - c[0])), (w * gC[1]) * (c[1] * (1.0f - c[1])), (w * gC[2]) * (c[2] * (1.0f - c[2]))};
        float4* dst = gr.sorted + (long long)pos * QE;
        if constexpr (KE == 1) {  // base-channel record: index quad + (density, degree-0 r, g, b) -- nothing left to expand
          store_f4<RF_NT_RECORD_STORE>(dst + 0, make_float4(ix[u][0], ix[u][1], ix[u][2], 0.0f));
          store_f4<RF_NT_RECORD_STORE>(dst + 1, make_float4(g_pre * g.rho, graw[0] * kC0, graw[1] * kC0, graw[2] * kC0));
        } else {  // compact specular record (48 B): the brick pass multiplies the SH basis of `vdir` in
          store_f4<RF_NT_RECORD_STORE>(dst + 0, make_float4(ix[u][0], ix[u][1], ix[u][2], g_pre * g.rho));
          store_f4<RF_NT_RECORD_STORE>(dst + 1, make_float4(graw[0], graw[1], graw[2], vdir[0]));
          store_f4<RF_NT_RECORD_STORE>(dst + 2, make_float4(vdir[1], vdir[2], 0.0f, 0.0f));
        }
      }
    }
  }
}

template <int K, bool DIFFUSE>
__global__ __launch_bounds__(kBlock, RF_EMIT_WAVES) void render_emit_direct_kernel(GridArgs g, RayArgs r, OutArgs fwd, GradArgs gr, uint32_t flags) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  render_emit_direct_ray<K, DIFFUSE>(g, r, fwd, gr, flags, (long long)blockIdx.x * kWavesPerBlock + wave, lane);
}

// the adjoints of both renders of a training iteration in one launch (waves 0, 1 of a block: specular, waves 2, 3: render_diffuse, of
// the same two rays): the kernel is bound by its per-ray dependency chains, and one launch of twice the waves has one tail instead of two
struct EmitPair {
  RayArgs r[2];
  OutArgs fwd[2];
  GradArgs gr[2];
  uint32_t flags[2];
};
template <int K>
__global__ __launch_bounds__(kBlock, RF_EMIT_WAVES) void render_emit_direct_pair_kernel(GridArgs g, EmitPair p) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long ray = (long long)blockIdx.x * (kWavesPerBlock / 2) + (wave & 1);
  // (the counters of BOTH forward passes are cleared by all threads of the grid here; the per-ray bodies get no clearing job)
  for (int i = 0; i < 2; ++i)
    if (p.gr[i].hist_clear)
      for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < p.gr[i].nkeys; k += (long long)gridDim.x * blockDim.x) p.gr[i].hist_clear[k] = 0;
  GradArgs gr = p.gr[wave < 2 ? 0 : 1];
  gr.hist_clear = nullptr;
  if (wave < 2) {
    if constexpr (K == 1)
      render_emit_direct_ray<1, true>(g, p.r[0], p.fwd[0], gr, p.flags[0], ray, lane);
    else
      render_emit_direct_ray<K, false>(g, p.r[0], p.fwd[0], gr, p.flags[0], ray, lane);
  } else {
    render_emit_direct_ray<1, true>(g, p.r[1], p.fwd[1], gr, p.flags[1], ray, lane);
  }
}

// =============================================================================================
// binned backward: the same adjoint without float atomics
//
// The atomic scatter above is pinned on the memory-side atomic unit (~21 G sector requests/s).  This path aggregates
// on chip first, under EXCLUSIVE ownership so that no atomics are needed at all:
//   1. render_backward_kernel<.., EMIT=true> writes one 32-byte gradient record per contributing sample and a 16-bit
//      key per slot: key = brick * 8 + flags, brick = the B^3-NODE brick holding the cell's lower node, flag bit a = the
//      cell's upper node on axis a lies in the next brick; it also counts the records per key;
//   2. bin_offsets_kernel + scatter_records_kernel (counting sort, atomic cursors) -- or torch.sort +
//      expand_records_kernel (stable, fixed summation order) -- put the EXPANDED records (index + per-channel values, SH
//      basis multiplied in) in key order;
//   3. brick_gather_kernel: one workgroup per brick reads the <= 14 key ranges that touch its nodes, sums them in MFMA
//      accumulators (weights x per-channel values, four records per instruction) and writes the brick with plain stores -- or
//      applies the optimizer step right there.
// History and measurements: DESIGN.md section 4.
// =============================================================================================
constexpr int kMaxListsPerKind = 8;   // lists of one kind per brick pass (data parallel: one per source rank)
constexpr int kRangeEntries = 15;     // contiguous key ranges of ONE sorted list that reach into a brick (brick_range_entry)
constexpr int kMaxRangesKind = kRangeEntries * kMaxListsPerKind;

struct BrickList {
  const float4* rec;         // sorted records: record_quads(K) float4 each
  const long long* offsets;  // [8 * nbricks + 1] start of every key class in `rec` (absolute positions)
};

// fused optimizer of the brick flush (torch.optim.Adam arithmetic, the same expressions as adam_kernel): the workgroup that
// owns a brick holds its complete gradient in LDS, so the update is applied there and the gradient never goes to HBM
// One Adam update (torch.optim.Adam's expressions: exp_avg, exp_avg_sq, denom = sqrt(exp_avg_sq) / sqrt(bias_correction2) + eps,
// param -= lr / bias_correction1 * exp_avg / denom), shared by rf_adam_step's kernel and the brick pass's optimizer flush so that the
// two stay bit-identical.  The square root and the reciprocal are the hardware's (v_sqrt_f32 / v_rcp_f32, 1 ulp): the IEEE
// sequences the compiler emits for sqrtf and '/' cost ~35 instructions per parameter and made the flush instruction-bound
// (14.8 K of its ~31 K cycles per brick without any memory traffic); 1 ulp of the update is 1e-7 x lr.
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float step, float b1, float b2, float eps,
                                            float inv_bc2_sqrt) {
  const float mm = m + (g - m) * (1.0f - b1);
  const float vv = v * b2 + (g * g) * (1.0f - b2);
  const float denom = __builtin_amdgcn_sqrtf(vv) * inv_bc2_sqrt + eps;
  p = p - step * (mm * __builtin_amdgcn_rcpf(denom));
  m = mm;
  v = vv;
}

struct AdamArgs {
  float* p1;  // parameters, first / second tensor (the tensors GridArgs describes, writable)
  float* p2;
  float* m1;  // exp_avg
  float* m2;
  float* v1;  // exp_avg_sq
  float* v2;
  float step;      // lr / (1 - beta1^t)
  float b1, b2, eps;
  float inv_bc2_sqrt;  // 1 / sqrt(1 - beta2^t)
  int byte_offsets_fit_32_bits;  // every element of the grid tensors lies below 2^30: the one-round flush addresses with 32-bit byte offsets
};

struct BrickArgs {
  BrickList wide[kMaxListsPerKind];    // full-width lists: compact K-records (kernel<K>), or base-channel records (kernel<1>)
  BrickList narrow[kMaxListsPerKind];  // kernel<K > 1> only: base-channel records of render_diffuse passes, summed into channels 0..3
  int nwide, nnarrow;
  int shift;               // log2(B)
  int nbx, nby, nbz;
  int brick_first;         // workgroup i handles brick brick_first + i (data parallel: the rank's own x-slabs)
  int accumulate;          // 0: grad = brick sum (no zero-fill needed), 1: grad += brick sum
  int fmul;                // reference layout: feature index of degree-0 colour c is c * fmul (base-only lists on an SH grid)
  int stagger;             // development builds (RF_BRICK_PROFILE / RF_BRICK_ABLATE): ablation switches from $RF_BRICK_STAGGER (0x100000 no tile
                           // loop, 0x200000 no lists, 0x400000 no flush)
  AdamArgs adam;           // only read by the ADAM instantiation
  // SPLIT launches (rf_brick_accumulate_adam_split): `parts` workgroups per brick, workgroup (brick, part) sums the lists l with
  // l % parts == part of each kind; the partial images meet in `partial`, the last workgroup of a brick to arrive adds them and flushes
  int parts;
  float* partial;          // [num_bricks][parts][brick_acc_words] partial accumulator images
  int* part_state;         // [num_bricks][1 + parts]: arrival counter, then one "wrote an image" flag per part; all zero between launches
};



// Sorted position p (>= *begin: the slots without gradient sort in front) receives the record of slot perm[p] (Q quads each;
// one thread per quad, coalesced 16-byte stores).
template <int Q>
__global__ void expand_records_kernel(const float4* __restrict__ rec, const long long* __restrict__ perm,
                                      const long long* __restrict__ begin_ptr, long long capacity, float4* __restrict__ out) {
  const long long begin = *begin_ptr;
  const long long items = (capacity - begin) * Q;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long long)gridDim.x * blockDim.x) {
    const long long i = begin + it / Q;
    const int part = (int)(it % Q);
    out[i * Q + part] = rec[perm[i] * Q + part];
  }
}

// Counting-sort alternative to torch.sort + expand: hist[k] = records per key (filled by the emit pass) ->
// offsets[k] = exclusive prefix sum (int64, what rf_brick_accumulate reads), cursor[k] = the same as int32 for the
// scatter pass; hist is cleared for the next iteration.  One workgroup (at most 32768 keys).
struct BinLists {  // up to two independent lists per launch (blockIdx.y)
  const int* hist[2];
  long long* offsets[2];
  int* cursor[2];
};

__device__ __forceinline__ void bin_offsets_body(const BinLists& lists, int nkeys, int list, int block) {
  // workgroup i owns keys [1024 i, 1024 i + 1024): it sums everything in front of its segment (coalesced, L2-resident)
  // and scans its own segment with wave shuffles -- no dependency between workgroups
  const int* __restrict__ hist = lists.hist[list];
  long long* __restrict__ offsets = lists.offsets[list];
  int* __restrict__ cursor = lists.cursor[list];
  __shared__ int s_front[16], s_own[16];
  const int t = threadIdx.x, lane = t & (kWave - 1), wave = t >> 6;
  const int seg0 = block * 1024;
  int front = 0;
  {  // (seg0 is a multiple of 1024 keys: whole 16-byte loads, four of them in flight per thread -- 64 segments at 128^3 / 4 x 8 x 8 bricks)
    const int4* __restrict__ h4 = reinterpret_cast<const int4*>(hist);
    const int n4 = seg0 >> 2;
    int k4 = t;
    for (; k4 + 3 * 1024 < n4; k4 += 4 * 1024) {
      const int4 a0 = h4[k4], a1 = h4[k4 + 1024], a2 = h4[k4 + 2048], a3 = h4[k4 + 3072];
      front += (a0.x + a0.y) + (a0.z + a0.w) + (a1.x + a1.y) + (a1.z + a1.w) + (a2.x + a2.y) + (a2.z + a2.w) + (a3.x + a3.y) + (a3.z + a3.w);
    }
    for (; k4 < n4; k4 += 1024) {
      const int4 a0 = h4[k4];
      front += (a0.x + a0.y) + (a0.z + a0.w);
    }
  }
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) front += __shfl_xor(front, d);
  const int k = seg0 + t;
  const int c = k < nkeys ? hist[k] : 0;
  int incl = c;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  if (lane == kWave - 1) {
    s_front[wave] = front;
    s_own[wave] = incl;
  }
  __syncthreads();
  int base = 0, before = 0, own = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    base += s_front[w];
    if (w < wave) before += s_own[w];
    own += s_own[w];
  }
  if (k < nkeys) {
    const int excl = base + before + incl - c;
    offsets[k] = excl;
    cursor[k] = excl;
  }
  if (t == 0 && seg0 + 1024 >= nkeys) offsets[nkeys] = base + own;
}

__global__ __launch_bounds__(1024) void bin_offsets_kernel(BinLists lists, int nkeys) { bin_offsets_body(lists, nkeys, blockIdx.y, blockIdx.x); }

// Every keyed slot takes the next free position of its key class (run-aggregated atomic cursor) and its expanded record
// is written there.  One wave per 256 consecutive slots (4 chunks of 64 whose atomics are in flight together): active
// slots are compacted into LDS, then groups of Q lanes write the Q quads of a record (coalesced 16-byte stores).  The
// order inside a class depends on the atomics' timing (unlike the sort path, results are not run-to-run bit-identical).
// Also clears `hist` for the next iteration.
template <int Q>
__global__ __launch_bounds__(256) void scatter_records_kernel(const short* __restrict__ keys, const float4* __restrict__ rec,
                                                              long long capacity, int* __restrict__ cursor,
                                                              float4* __restrict__ out, int* __restrict__ hist, int nkeys) {
  constexpr int NCH = 4;  // chunks per wave
  __shared__ long long s_slot[4][NCH * kWave];
  __shared__ int s_pos[4][NCH * kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
  if (hist)
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < nkeys; k += (long long)gridDim.x * blockDim.x) hist[k] = 0;
  const long long slot0 = ((long long)blockIdx.x * 4 + wave) * (NCH * kWave);
  if (slot0 >= capacity) return;
  int key[NCH], run[NCH], hl[NCH], base[NCH];
  bool head[NCH];
  unsigned long long act[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const long long slot = slot0 + i * kWave + lane;
    key[i] = slot < capacity ? (int)keys[slot] : -1;
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const bool active = key[i] >= 0;
    act[i] = __ballot(active);
    const int prev = __shfl_up(key[i], 1);
    head[i] = active && (lane == 0 || prev != key[i]);
    const unsigned long long heads = __ballot(head[i]);
    const unsigned long long ends = __ballot(head[i] || !active);
    const unsigned long long above = (lane == 63) ? 0ull : (ends & ~((2ull << lane) - 1ull));
    run[i] = (above ? __builtin_ctzll(above) : 64) - lane;
    const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    hl[i] = below ? 63 - __builtin_clzll(below) : 0;
  }
#pragma unroll
  for (int i = 0; i < NCH; ++i) base[i] = head[i] ? atomicAdd(&cursor[key[i]], run[i]) : 0;  // 4 atomics in flight
  int n_act = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int pos = __shfl(base[i], hl[i]) + (lane - hl[i]);
    if (key[i] >= 0) {
      const int e = n_act + __popcll(act[i] & ((1ull << lane) - 1ull));
      s_slot[wave][e] = slot0 + i * kWave + lane;
      s_pos[wave][e] = pos;
    }
    n_act += __popcll(act[i]);
  }
  wave_lds_fence();
  const int items = n_act * Q;
  for (int it = lane; it < items; it += kWave) {
    const int e = it / Q, part = it - e * Q;
    out[(long long)s_pos[wave][e] * Q + part] = rec[s_slot[wave][e] * Q + part];
  }
}

// One workgroup (8 waves) per NODE brick: it owns B^3 nodes exclusively, so the result is written with plain stores (or consumed
// by the optimizer on the spot).  Records are sorted by key = cell-brick * 8 + flags, so the records of a source brick that reach
// into this brick (flags superset of the offset) are a handful of contiguous ranges: no record is read that does not contribute.
// The kernel is brick_gather_kernel below.  Its round-1 / early round-2 predecessor summed in LDS with read-add-writes under wave
// ownership of channel pairs (LDS float atomics retire 0.33 lane/clk/CU, tools/lds_microbench*.hip): ~200 cycles per pair of records
// on the read-add-write chain plus float64 LDS atomics for the base-channel records, 0.489 ms on the bench step against 0.394 now.
constexpr int kBrickThreads = 512;  // 8 waves; LDS admits 2 workgroups per CU

// geometry of the accumulator image the flush reads: node stride CS = channels rounded up to a multiple of 4 (float4 flush);
// rows (z runs) and slabs (x) are padded to odd multiples of 16 / 8 words (bank spread of the image writes and the flush reads)
__host__ __device__ inline int brick_node_stride(int C) { return (C + 3) / 4 * 4; }
__host__ __device__ inline int brick_row_stride(int B, int C) {
  const int row = B * brick_node_stride(C);
  return row + ((48 - row % 64) + 64) % 64;
}
__host__ __device__ inline int brick_slab_stride(int B, int C) {
  const int slab = B * brick_row_stride(B, C);
  return slab + ((8 - slab % 64) + 64) % 64;
}
__host__ __device__ inline int brick_acc_words(int B, int C) { return B * brick_slab_stride(B, C) + 64; }  // + trash row

// Phase timing of the brick pass (development builds only: -DRF_BRICK_PROFILE; tools/brick_phase_profile.py): thread 0 of
// every workgroup adds the s_memtime span of each phase to a global table.
#ifdef RF_BRICK_PROFILE
__device__ unsigned long long g_brick_prof[8];
// (the spans are summed in registers and added to the table once, at the end of the kernel: an atomic per mark is a memory
// operation the next vmcnt wait has to sit out, which tripled the kernel time and mis-attributed it)
#define RF_PROF_MARK(slot)                                                       \
  do {                                                                           \
    const unsigned long long now_ = __builtin_readcyclecounter();                \
    prof_acc_[slot] += now_ - prof_t_;                                           \
    prof_t_ = now_;                                                              \
  } while (0)
#define RF_PROF_START()                                              \
  unsigned long long prof_acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0};        \
  unsigned long long prof_t_ = __builtin_readcyclecounter()
#define RF_PROF_END()                                                            \
  do {                                                                           \
    if (threadIdx.x == 0) {                                                      \
      _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_)                           \
        if (prof_acc_[s_]) atomicAdd(&g_brick_prof[s_], prof_acc_[s_]);          \
    }                                                                            \
  } while (0)
#else
#define RF_PROF_MARK(slot) do { } while (0)
#define RF_PROF_START() do { } while (0)
#define RF_PROF_END() do { } while (0)
#endif

// Which ranges of a sorted list reach into brick (bx, by, bz).  A record with key (source brick, f_x, f_y, f_z) touches the bricks
// source + o for every offset o <= f (component-wise), so brick b receives, from the source brick b - o, the classes f >= o.  In the
// key order of brick_key -- ((2 sx + f_x) * nby + sy) * nbz + sz) * 4 + (f_y | 2 f_z) -- those are, per (o_x, f_x) in
// {(0,0), (0,1), (1,1)} and per (o_y, o_z): o_yz = (0,0): classes 0..3; (1,0): {1}, {3}; (0,1): 2..3; (1,1): {3} -- 3 x 5 = 15
// contiguous ranges.  Entry e = 5 e_x + e_yz.
__device__ __forceinline__ void brick_range_entry(const BrickArgs& a, const long long* __restrict__ offsets, int bx, int by, int bz, int e,
                                                  int& start, int& cnt) {
  const int ex = e / 5, eyz = e - ex * 5;
  const int ox = ex == 2, xf = ex >= 1;
  const int oy = (0x16 >> eyz) & 1, oz = (0x18 >> eyz) & 1;
  const int f0 = (0x32310 >> (4 * eyz)) & 7, f1 = (0x33313 >> (4 * eyz)) & 7;
  const int sx = bx - ox, sy = by - oy, sz = bz - oz;
  start = 0;
  cnt = 0;
  if (sx >= 0 && sy >= 0 && sz >= 0) {
    const long long* off = offsets + ((long long)((((sx << 1) | xf) * a.nby + sy) * a.nbz + sz) << 2);
    const long long rs = off[f0];
    start = (int)rs;
    cnt = (int)(off[f1 + 1] - rs);
  }
}

// The last phase of a brick workgroup: the sums of the B^3 owned nodes (LDS accumulators `acc`, node (x, y, z) channel c at
// x * SX + y * SY + z * CS + c; all zero when `any` is false) go out with plain stores -- or, ADAM, are consumed by the
// optimizer step on the spot.
// MIRROR (ADAM, ONE_ROUND, 4 x 8 x 8 bricks, grid dims multiples of the brick): the updated parameters ALSO go out in the reference's
// own layout -- densities [X,Y,Z,1] at `gdens`, features [X,Y,Z,3K] (index = colour * K + k, process.py:61,66) at `gfeat` -- so that the
// Parameters of a reference-storage grid whose split shadow this pass updates stay in sync without a re-layout launch (81 us per
// iteration of the strict drop-in step: 235 MB read + 235 MB written; here 235 MB written out of LDS).  Every thread parks its
// updated quads in the accumulator image, in place of the gradient quads it alone consumed; a z column of 8 nodes is then 8 * 3K
// consecutive floats of the feature tensor = 2 * 3K whole 16-byte stores (the column starts at a multiple of 8 nodes).
template <int K, bool ADAM, bool ONE_ROUND, int TH = kBrickThreads, int BX = 8, bool MIRROR = false>
__device__ __forceinline__ void brick_flush(const GridArgs& g, const BrickArgs& a, const float* acc, bool any, int X0, int Y0, int Z0,
                                            float* gdens, float* gfeat) {
  static_assert(!MIRROR || (ADAM && ONE_ROUND && BX == 4), "the mirror write-out rides on the one-round optimizer flush of 4 x 8 x 8 bricks");
  constexpr int C = 3 * K + 1;
  constexpr int CS = (C + 3) / 4 * 4;
  const int B = ONE_ROUND ? 8 : (1 << a.shift);  // (ONE_ROUND: the host launches it for 8^3 bricks only -- strides fold to constants)
  const int SY = brick_row_stride(B, C), SX = brick_slab_stride(B, C);
  const int tid = threadIdx.x;
  // ---- write the brick out: plain stores (exclusive owner), contiguous runs along z
  const bool split = g.layout == RF_LAYOUT_SPLIT;
  if (split && (C & 3) == 0 && (g.dstride & 3) == 0 && (C == 4 || (g.fstride & 3) == 0)) {
    // split layout, whole float4s: base [X,Y,Z,4] = channels 0..3 of a node, rest [X,Y,Z,C-4] = channels 4..C-1
    constexpr int QN = C / 4;  // float4s per node
    constexpr int QR = QN > 1 ? QN - 1 : 1;
    const int nq = (BX == 4 ? 4 : B) * B * B * QN;  // (BX = 4: bricks of 4 x 8 x 8 nodes -- four x columns)
    // i -> (column (x, y), quad group, z, quad) with the quads of one tensor contiguous along z
    auto quad_of = [&](int i, int& fx, int& fy, int& fz, int& qd) -> bool {
      const int col = (i >> a.shift) / QN, r = i - col * (B * QN);  // (division by a constant)
      const bool first = r < B;  // the B base quads of the column come first
      fz = first ? r : (r - B) / QR;
      qd = first ? 0 : 1 + (r - B) - fz * QR;
      fx = col >> a.shift;
      fy = col & (B - 1);
      return i < nq && X0 + fx < g.X && Y0 + fy < g.Y && Z0 + fz < g.Z;
    };
    if constexpr (ADAM) {
      // The optimizer step on the parameters this workgroup holds the complete gradient of (adam_kernel's expressions):
      // parameters with ordinary accesses (the next forward pass reads them), moments streamed non-temporally.  The flush is
      // the only phase of the kernel that waits on HBM: a thread issues the 3 x U loads of U quads before it touches the
      // first one (one quad at a time kept ~24 KB per CU in flight and ran at a third of the HBM rate).
      // Element offsets are kept as 32-bit (host-checked) to stay inside the 128-register budget of 4 waves per SIMD.
      // vmcnt counts loads and stores together and does not order them against each other, so a second round's first wait also sits
      // out the first round's stores: the flush was two dependent memory round trips (4 + 3 quads per thread; 0.414 ms for the
      // pass).  ALL 21 loads of a thread in flight before its first store needs 84 data registers -- and fits the 128-register budget
      // only when the addresses cost next to nothing: round 2 tried it with per-lane tensor selection (64-bit addresses per
      // load: spills, 0.419 ms); the quad -> thread assignment below makes every instruction address ONE tensor: 0.366 ms.
      constexpr int U = (QN == 7) ? 4 : 1;
      const AdamArgs& ad = a.adam;
      if constexpr (ONE_ROUND) {  // (host: bricks of 8^3 nodes, every element of the grid tensors below 2^30)
        // ONE round: thread t takes the base quad of node t (512 nodes) and rest quads t, t + 512, ... of the brick's 3072 (4 x 8 x 8
        // bricks, TH = 256: of node t of 256 and rest quads t, t + 256, ... of 1536: the same seven quads per thread) -- so that
        // every load / store instruction of a wave addresses ONE tensor: uniform base (SGPR pair) + a 32-bit byte offset per lane,
        // the same offset for parameter, exp_avg and exp_avg_sq.  7 offsets + 84 data registers: all 21 loads of a thread are in
        // flight before its first store (vmcnt counts loads and stores together, so a second round's first wait would also sit out
        // the first round's stores).  (Byte offsets in 32 bits: the host checks the tensors for < 2^30 elements.)
        unsigned int bo[QN];  // byte offset inside the quad's tensor; 0xffffffff: nothing to do
        int lds_at[QN];
        float4 p4[QN];
        vf4 m4[QN], v4[QN];
#pragma unroll
        for (int u = 0; u < QN; ++u) {
          int fx, fy, fz, qd;
          if (u == 0) {
            fx = tid >> 6;
            fy = (tid >> 3) & 7;
            fz = tid & 7;
            qd = 0;
          } else {
            const int j = (u - 1) * TH + tid, col = j / 48, r = j - col * 48;
            fz = r / 6;
            qd = 1 + r - 6 * fz;
            fx = col >> 3;
            fy = col & 7;
          }
          const bool ok = X0 + fx < g.X && Y0 + fy < g.Y && Z0 + fz < g.Z;
          const unsigned int lin = ok ? node_lin(g, X0 + fx, Y0 + fy, Z0 + fz) : 0u;
          // (24-bit multiplies: the host launches this path for <= 2^24 nodes and strides < 2^24 only)
          const unsigned int o = u == 0 ? __umul24(lin, (unsigned)g.dstride) : __umul24(lin, (unsigned)g.fstride) + 4u * (unsigned)(qd - 1);
          bo[u] = ok ? o * 4u : 0xffffffffu;
          lds_at[u] = fx * SX + fy * SY + fz * CS + 4 * qd;
        }
#pragma unroll
        for (int u = 0; u < QN; ++u) {
          if (bo[u] != 0xffffffffu) {
            const char* pb = reinterpret_cast<const char*>(u == 0 ? ad.p1 : ad.p2);
            const char* mb = reinterpret_cast<const char*>(u == 0 ? ad.m1 : ad.m2);
            const char* vb = reinterpret_cast<const char*>(u == 0 ? ad.v1 : ad.v2);
            p4[u] = *reinterpret_cast<const float4*>(pb + bo[u]);
            m4[u] = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(mb + bo[u]));
            v4[u] = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(vb + bo[u]));
          }
        }
#pragma unroll
        for (int u = 0; u < QN; ++u) {
          if (bo[u] == 0xffffffffu) continue;
          const float4 gq = any ? *reinterpret_cast<const float4*>(&acc[lds_at[u]]) : make_float4(0.f, 0.f, 0.f, 0.f);
          float gg[4] = {gq.x, gq.y, gq.z, gq.w};
          float pn[4] = {p4[u].x, p4[u].y, p4[u].z, p4[u].w};
          if (u == 0 && g.mode == RF_DENSITY_ABS) {  // d|x|/dx of the raw density (the parameter before its update)
            const float dv = pn[0] * g.rho;
            gg[0] = (dv > 0.f) ? gg[0] : ((dv < 0.f) ? -gg[0] : 0.0f);
          }
          vf4 mn, vn;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float mm = m4[u][c], vv = v4[u][c];
            adam_update(pn[c], mm, vv, gg[c], ad.step, ad.b1, ad.b2, ad.eps, ad.inv_bc2_sqrt);
            mn[c] = mm;
            vn[c] = vv;
          }
          char* pb = reinterpret_cast<char*>(u == 0 ? ad.p1 : ad.p2);
          char* mb = reinterpret_cast<char*>(u == 0 ? ad.m1 : ad.m2);
          char* vb = reinterpret_cast<char*>(u == 0 ? ad.v1 : ad.v2);
          *reinterpret_cast<float4*>(pb + bo[u]) = make_float4(pn[0], pn[1], pn[2], pn[3]);
          __builtin_nontemporal_store(mn, reinterpret_cast<vf4*>(mb + bo[u]));
          __builtin_nontemporal_store(vn, reinterpret_cast<vf4*>(vb + bo[u]));
          if constexpr (MIRROR) *reinterpret_cast<float4*>(const_cast<float*>(&acc[lds_at[u]])) = make_float4(pn[0], pn[1], pn[2], pn[3]);
        }
        if constexpr (MIRROR) {
          __syncthreads();
          constexpr int F = 3 * K, FQ = 2 * F, CQ = FQ + 2;  // float4s of a column of 8 nodes: features, + 2 of densities
          for (int i = tid; i < 32 * CQ; i += TH) {
            const int col = i / CQ, r = i - col * CQ;
            const int fx = col >> 3, fy = col & 7;
            const float* img = acc + fx * SX + fy * SY;
            const unsigned int lin0 = node_lin(g, X0 + fx, Y0 + fy, Z0);
            float v[4];
            if (r < FQ) {
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int e = 4 * r + t;
                const int nz = e / F, f = e - nz * F;
                const int colour = f / K, k = f - colour * K;
                v[t] = img[nz * CS + (k == 0 ? 1 + colour : 4 + colour * (K - 1) + (k - 1))];
              }
              *reinterpret_cast<float4*>(gfeat + (size_t)(__umul24(lin0, (unsigned)F) + 4u * (unsigned)r)) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
              const int z0 = 4 * (r - FQ);
#pragma unroll
              for (int t = 0; t < 4; ++t) v[t] = img[(z0 + t) * CS];
              *reinterpret_cast<float4*>(gdens + (size_t)(lin0 + (unsigned)z0)) = make_float4(v[0], v[1], v[2], v[3]);
            }
          }
        }
        return;
      } else {
      // (bricks of 4^3 nodes, tensors of 2^30 elements and more: two rounds of 4 + 3 quads per thread with per-lane tensor selection)
      for (int i0 = tid; i0 < nq; i0 += TH * U) {
        unsigned int off[U];  // bit 31: rest tensor; 0xffffffff: nothing to do
        float4 p4[U];
        vf4 m4[U], v4[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          int fx, fy, fz, qd;
          const bool ok = quad_of(i0 + u * TH, fx, fy, fz, qd);
          const unsigned int lin = ok ? node_lin(g, X0 + fx, Y0 + fy, Z0 + fz) : 0u;
          const unsigned int o = qd == 0 ? lin * (unsigned)g.dstride : (lin * (unsigned)g.fstride + 4u * (unsigned)(qd - 1)) | 0x80000000u;
          off[u] = ok ? o : 0xffffffffu;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (off[u] != 0xffffffffu) {
            const bool rest = off[u] >> 31;
            const unsigned int o = off[u] & 0x7fffffffu;
#ifdef RF_BRICK_PROFILE
            if (a.stagger & 0x10000) { p4[u] = make_float4(0.f, 0.f, 0.f, 0.f); m4[u] = vf4{0.f, 0.f, 0.f, 0.f}; v4[u] = vf4{1.f, 1.f, 1.f, 1.f}; continue; }
#endif
            p4[u] = *reinterpret_cast<const float4*>((rest ? ad.p2 : ad.p1) + o);
            m4[u] = __builtin_nontemporal_load(reinterpret_cast<const vf4*>((rest ? ad.m2 : ad.m1) + o));
            v4[u] = __builtin_nontemporal_load(reinterpret_cast<const vf4*>((rest ? ad.v2 : ad.v1) + o));
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (off[u] == 0xffffffffu) continue;
          const bool rest = off[u] >> 31;
          const unsigned int o = off[u] & 0x7fffffffu;
          int fx, fy, fz, qd;
          quad_of(i0 + u * TH, fx, fy, fz, qd);
          const float4 gq = any ? *reinterpret_cast<const float4*>(&acc[fx * SX + fy * SY + fz * CS + 4 * qd]) : make_float4(0.f, 0.f, 0.f, 0.f);
          float gg[4] = {gq.x, gq.y, gq.z, gq.w};
          float pn[4] = {p4[u].x, p4[u].y, p4[u].z, p4[u].w};
          if (!rest && g.mode == RF_DENSITY_ABS) {  // d|x|/dx of the raw density (the parameter before its update)
            const float dv = pn[0] * g.rho;
            gg[0] = (dv > 0.f) ? gg[0] : ((dv < 0.f) ? -gg[0] : 0.0f);
          }
          vf4 mn, vn;
#ifdef RF_BRICK_PROFILE
          if (a.stagger & 0x40000) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { mn[c] = m4[u][c] + gg[c]; vn[c] = v4[u][c]; pn[c] += gg[c]; }
          } else
#endif
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float mm = m4[u][c], vv = v4[u][c];
            adam_update(pn[c], mm, vv, gg[c], ad.step, ad.b1, ad.b2, ad.eps, ad.inv_bc2_sqrt);
            mn[c] = mm;
            vn[c] = vv;
          }
#ifdef RF_BRICK_PROFILE
          if (a.stagger & 0x20000) { if (pn[0] == 123.456f) *reinterpret_cast<float4*>((rest ? ad.p2 : ad.p1) + o) = make_float4(mn[0], vn[1], pn[2], pn[3]); continue; }
#endif
          *reinterpret_cast<float4*>((rest ? ad.p2 : ad.p1) + o) = make_float4(pn[0], pn[1], pn[2], pn[3]);
          __builtin_nontemporal_store(mn, reinterpret_cast<vf4*>((rest ? ad.m2 : ad.m1) + o));
          __builtin_nontemporal_store(vn, reinterpret_cast<vf4*>((rest ? ad.v2 : ad.v1) + o));
        }
      }
      }
      return;
    }
    for (int i = tid; i < nq; i += TH) {
      int fx, fy, fz, qd;
      if (!quad_of(i, fx, fy, fz, qd)) continue;
      const long long lin = node_lin(g, X0 + fx, Y0 + fy, Z0 + fz);
      float4 v = any ? *reinterpret_cast<const float4*>(&acc[fx * SX + fy * SY + fz * CS + 4 * qd]) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (qd == 0 && g.mode == RF_DENSITY_ABS) {  // d|x|/dx of the raw density, applied once per node
        const float dv = g.dens[lin * g.dstride] * g.rho;
        v.x = (dv > 0.f) ? v.x : ((dv < 0.f) ? -v.x : 0.0f);
      }
      const long long off = (qd == 0) ? lin * g.dstride : lin * g.fstride + 4 * (qd - 1);
      if (qd == 0) {
        float4* dst = reinterpret_cast<float4*>(gdens + off);
        if (a.accumulate) {
          const float4 o = *dst;
          v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
        }
        *dst = v;
      } else {
        float4* dst = reinterpret_cast<float4*>(gfeat + off);
        if (a.accumulate) {
          const float4 o = *dst;
          v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
        }
        // non-temporal: nothing reads the `rest` gradients again before the optimizer (or the collective) streams them
        const vf4 v4 = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(v4, reinterpret_cast<vf4*>(dst));
      }
    }
    return;
  }
  const int n_first = split ? 4 : 1;  // channels of a node that live in the densities/base tensor
  const int n_second = C - n_first;
  for (int pass = 0; pass < 2; ++pass) {
    const int nch = pass == 0 ? n_first : n_second;
    if (nch == 0) continue;
    const int run = B * nch;  // floats of one z column in this tensor
    float* out = pass == 0 ? gdens : gfeat;
    const long long ostride = pass == 0 ? g.dstride : g.fstride;
    for (int i = tid; i < (BX == 4 ? 4 : B) * B * run; i += TH) {
      const int col = i / run, r = i - col * run;
      const int fz = r / nch, c2 = r - fz * nch;
      const int fx = col >> a.shift, fy = col & (B - 1);
      const int X = X0 + fx, Y = Y0 + fy, Z = Z0 + fz;
      if (X >= g.X || Y >= g.Y || Z >= g.Z) continue;
      int lds_c;
      if (split) {
        lds_c = pass * 4 + c2;
      } else if (pass == 0) {
        lds_c = 0;
      } else {
        const int col3 = c2 / K, kk = c2 - col3 * K;  // reference order colour * K + k
        lds_c = (kk == 0) ? 1 + col3 : 4 + col3 * (K - 1) + (kk - 1);
      }
      const long long lin = node_lin(g, X, Y, Z);
      float v = any ? acc[fx * SX + fy * SY + fz * CS + lds_c] : 0.0f;
      if (lds_c == 0 && g.mode == RF_DENSITY_ABS) {  // d|x|/dx of the raw density, applied once per node
        const float dv = g.dens[lin * g.dstride] * g.rho;
        v = (dv > 0.f) ? v : ((dv < 0.f) ? -v : 0.0f);
      }
      float* dst = out + lin * ostride + (long long)c2 * (pass == 1 && !split ? a.fmul : 1);
      *dst = a.accumulate ? (*dst + v) : v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// brick_gather_kernel: one workgroup per brick of B^3 owned nodes; the sums live in MFMA accumulators.
//
// The gradient of a brick is  acc[node][channel] = sum over records r of  W[node][r] * G[r][channel],  where column r of W holds
// the 8 trilinear weights of record r (zero for the other nodes) and row r of G the record's per-channel values: a sparse x dense
// product.  (Summing in LDS with read-add-writes cost ~200 cycles per pair of records on the dependent chain.)  The owned
// nodes are cut into tiles of 16 nodes (2 x 2 x 4), a batch of records (256, staged in LDS) is binned to the tiles it touches
// (2.8 on average, ballots: no atomics, list order = record order, so the summation order is fixed by the record order), and a
// wave multiplies the records of a tile four at a time into the tile's 16 x 16 accumulator blocks with v_mfma_f32_16x16x4_f32
// (f32 in, f32 accumulate = an fmaf chain; 32 cycles per instruction).  A operand = weight of (node of the lane, record k): the
// product of three table entries the binning pass prepared per record and axis (weight at local node coordinate 0..7, mostly
// zero); B operand = channel value of record k.  Everything a wave needs per instruction is three 4-byte LDS reads for A and one
// per 16 channels for B; no LDS writes, no ownership rules, no serialisation of records that share nodes.
// Each wave owns 4 tiles (B = 8: 32 tiles), i.e. 4 x NT accumulator blocks of 4 registers.
// The MFMA is used as a scatter-add engine, not to make the pass compute-bound: the kernel stays HBM-bound by its flush
// (optimizer state) and record reads; this only removes the LDS chain that kept it from that bound.
// Phase times per brick on the bench step (development build, ~1.65 GHz under this load; 5 batches): flush 26 K cycles (9 K of them
// instructions), lists 8 K + tile loops 15 K (+ 6 K of barrier imbalance behind them), waiting for the first batch's loads 12 K,
// issuing loads 6 K, record pass 5 K, range set-up 3.5 K, image 4 K.  Tried and dropped: PERSISTENT workgroups that request the
// next brick's first batch before flushing (0.394 -> 0.53 ms: vmcnt counts loads and stores in issue order, so the next brick's
// first wait also sits out the completion of the whole flush's stores); 7 instead of 4 quads in flight in the flush (spills);
// an XCD-contiguous brick order (boundary records re-read through one L2: 0.394 -> 0.41 ms).
// ---------------------------------------------------------------------------------------------
// LDS reads the compiler cannot re-schedule (brick_gather_kernel's tile loop): the optimizer proves plain LDS loads re-computable and
// rotates a hand-pipelined loop back into read -> wait -> read -> wait -> multiply.  These are issued where they stand; the value
// may only be used after lds_wait() was given the same variable.  (LDS returns data in order, so the compiler's own lgkmcnt waits
// stay conservative in the presence of reads it does not know about.)
__device__ __forceinline__ uint32_t lds_offset(const void* p) { return (uint32_t)(uintptr_t)p; }
// (the destination is an in/out operand: the request lands in the variable's own register, and a variable is never copied
// between its request and its wait -- a copy would read the register before the data arrives)
template <int OFF>
__device__ __forceinline__ void lds_request_f32(float& v, uint32_t addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "+v"(v) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void lds_request_u8(uint32_t& v, uint32_t addr) {
  asm volatile("ds_read_u8 %0, %1" : "+v"(v) : "v"(addr));
}
__device__ __forceinline__ void lds_wait(uint32_t& a, uint32_t& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void lds_wait(float& a, float& b, float& c, float& d, float& e, uint32_t& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
__device__ __forceinline__ void lds_wait(float& a, float& b, float& c, float& d, float& e, float& e2, float& e3, uint32_t& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(e2), "+v"(e3), "+v"(f));
}

// tile tl (0..3) of wave `wave` = tl * (waves of the workgroup) + wave: interleaved over the waves (a wave's four tiles are spread along
// x), so that a hot corner of the brick does not land on one wave (0.414 -> 0.409 ms against four consecutive tiles per wave)
constexpr int kGatherBatch = 256;  // records per batch (their indices travel as bytes)
constexpr int kGatherWtab = 24;    // rows of the weight table: 3 axes x local node coordinate 0..7
// Bank conflicts are what bounds the tile loop (6 LDS reads per instruction): table rows are kGatherBatch + 4 words apart, so that
// the 2 (x, y) or 4 (z) rows the lanes of one record read, and those of the neighbouring record indices of the same instruction,
// fall into different banks (GROW in the kernel); record rows are padded by two quads (36 / 12 words), so that the four records of an
// instruction do not all start in the same two bank groups.
// words of a record's row of per-channel values in LDS (C4 = channels rounded up to whole quads)
__host__ __device__ constexpr int gather_record_words(int C4) { return C4 + 8; }
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (BX = 4: the 4 x 8 x 8 bricks of 256-thread workgroups -- batches of 128 records, an image of 4 slabs)
__host__ __device__ inline int gather_lds_words(int B, int C, int BX = 8) {
  const int C4 = (C + 3) / 4 * 4;
  const int gb = BX == 4 ? kGatherBatch / 2 : kGatherBatch;
  const int batch = gb * gather_record_words(C4) + (gb + 4) * kGatherWtab;  // per-channel rows, weight table
  const int image = BX == 4 ? 4 * brick_slab_stride(B, C) + 64 : brick_acc_words(B, C);
  return batch > image ? batch : image;
}

// (SH degree 3: 49 channels = four 16-channel blocks, 64 accumulator registers and 123 KB of LDS -- one workgroup per CU, which
// leaves a wave 256 registers)
// ONE_ROUND: the launch is for 8^3-node bricks (the brick edge folds to a constant); with ADAM it also selects the one-round flush
// (which additionally needs 32-bit byte offsets: the host checks)
// BX = 4 (RF_BRICK_4X8X8, the single-GPU optimizer pass): bricks of 4 x 8 x 8 nodes, summed by 256-thread workgroups in batches of 128
// records -- 16 tiles, four per wave as in the 8^3 case, half the LDS: FOUR workgroups per CU instead of two.  The pass is bound by the
// latencies of a workgroup's serial phases (ranges -> first records -> tile loops -> flush), and twice the workgroups hide twice as
// many of them; the price is one more brick face across x (a record is read 1.58 x instead of 1.42 x) and twice the keys.
template <int K, bool ADAM, bool ONE_ROUND = false, bool SPLIT = false, int BX = 8, bool MIRROR = false>
__global__ __launch_bounds__(BX == 4 ? kBrickThreads / 2 : kBrickThreads, (K > 9 ? 2 : 4)) void brick_gather_kernel(GridArgs g, BrickArgs a, float* gdens, float* gfeat) {
  static_assert(BX == 8 || (BX == 4 && ONE_ROUND && !SPLIT), "4 x 8 x 8 bricks: 8-node y and z edges, one workgroup per brick");
  constexpr int TH = BX == 4 ? kBrickThreads / 2 : kBrickThreads;  // threads of the workgroup
  constexpr int NWV = TH / 64;                                     // its waves: four tiles each
  constexpr int GB = TH / 2;                                       // records per batch (two threads per record)
  constexpr int GROW = GB + 4;                                     // words of a weight-table row (GROW)
  constexpr int C = 3 * K + 1;
  constexpr int C4 = (C + 3) / 4 * 4;
  constexpr int QW = record_quads(K);      // quads of a full-width record in HBM
  constexpr int QN = record_quads(1);      // quads of a base-channel (render_diffuse) record
  constexpr int NT = (C4 + 15) / 16;       // 16-channel blocks per tile
  static_assert(NT == 1 || NT == 2 || NT == 4, "one, two or four 16-channel blocks (SH degree 0 / base lists, 1-2, 3)");
  constexpr int CS = C4;
  constexpr int NW = GB / 64;    // mask words per tile
  extern __shared__ __attribute__((aligned(16))) float acc[];  // first the batch buffers, in the end the accumulator image the flush reads
  float* rows = acc;                                            // [GB][RW] per-channel values of the batch's records
  float* wtab = acc + GB * gather_record_words(C4);   // [3][8][GROW]: axis, local node coordinate, record
  __shared__ uint32_t s_tmask[GB];                   // record -> bit t: it touches tile t
  __shared__ unsigned char s_list[4 * NWV][GB + 24];      // tile -> its records (+ padding of the last instructions, + read-ahead slack)
  __shared__ __attribute__((aligned(16))) int s_wstart[kMaxRangesKind], s_wcum[kMaxRangesKind + 8];  // ranges of the full-width lists: first record, running count
  __shared__ __attribute__((aligned(16))) int s_nstart[kMaxRangesKind], s_ncum[kMaxRangesKind + 8];  // ... of the base-channel lists of a mixed call
  __shared__ int s_part[4];
  const int B = ONE_ROUND ? 8 : (1 << a.shift);  // (ONE_ROUND: 8^3 bricks only -- tile counts and image strides fold to constants)
  const int bshift = ONE_ROUND ? 3 : a.shift;
  const int SY = brick_row_stride(B, C), SX = brick_slab_stride(B, C);

  RF_PROF_START();
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const int wave = tid >> 6;
  // SPLIT: the `parts` workgroups of a brick are neighbours in the launch (they run side by side)
  const int parts = SPLIT ? a.parts : 1;
  const int part = SPLIT ? (int)(blockIdx.x % (unsigned)parts) : 0;
  const int brick_local = SPLIT ? (int)(blockIdx.x / (unsigned)parts) : (int)blockIdx.x;
  const int brick = a.brick_first + brick_local;
  // the lists of a kind this workgroup sums: l = part + ll * parts, ll < lists_of(nl)
  auto lists_of = [&](int nl) { return SPLIT ? (nl > part ? (nl - part + parts - 1) / parts : 0) : nl; };
  auto list_index = [&](int ll) { return SPLIT ? part + ll * parts : ll; };
  const int nwide_p = lists_of(a.nwide), nnarrow_p = lists_of(a.nnarrow);
  const int bz = brick % a.nbz, by = (brick / a.nbz) % a.nby, bx = brick / (a.nbz * a.nby);
  const int X0 = bx << (BX == 4 ? 2 : bshift), Y0 = by << bshift, Z0 = bz << bshift;
  // ---- range set-up: waves 0, 1 = the 15 ranges of each full-width list, waves 2, 3 = of each base-channel list; running
  // counts by wave scan (empty ranges stay in the tables with zero length: the record -> range walk skips them)
  {
    const int kind = tid >> 7, i = tid & 127;
    int start = 0, cnt = 0;
    if (tid < 256) {
      const int nl = kind ? nnarrow_p : nwide_p;
      if (i < kRangeEntries * nl) {
        const int l = nl == 1 ? 0 : i / kRangeEntries;
        const int li = list_index(l);
        const long long* offs = (!SPLIT && nl == 1) ? (kind ? a.narrow[0].offsets : a.wide[0].offsets) : (kind ? a.narrow[li].offsets : a.wide[li].offsets);
        brick_range_entry(a, offs, bx, by, bz, i - l * kRangeEntries, start, cnt);
      }
    }
    int cum = cnt;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const int up = __shfl_up(cum, d);
      if (lane >= d) cum += up;
    }
    if (wave < 4 && lane == kWave - 1) s_part[wave] = cum;
    __syncthreads();
    if (tid < 256) {
      if (wave & 1) cum += s_part[wave - 1];
      int* st_ = kind ? s_nstart : s_wstart;
      int* cu_ = kind ? s_ncum : s_wcum;
      if (i < kMaxRangesKind) {
        st_[i] = start;
        cu_[i + 1] = cum;
      }
      if (i == 0) cu_[0] = 0;
    }
    __syncthreads();
  }
  const int total = s_wcum[kMaxRangesKind];
  const int total_d = s_ncum[kMaxRangesKind];
  bool any = total > 0 || total_d > 0;
  if (!SPLIT && !any && a.accumulate) return;  // nothing reaches this brick
  RF_PROF_MARK(0);  // range set-up

  // tiles: 2 x 2 x 4 nodes; tile t = (px * npy + py) * npz + pz
  const int npz = B >> 2, npy = B >> 1;
  const int ntiles = BX == 4 ? 16 : (B * B * B) >> 4;
  const int mi = lane & 15;  // node of this lane inside a tile (A operand row / accumulator row group)
  const int mdx = mi >> 3, mdy = (mi >> 2) & 1, mdz = mi & 3;
  const int kk = lane >> 4;  // which of the 4 records of an instruction this lane feeds
  const int jj = lane & 15;  // channel inside a 16-channel block (B operand column / accumulator column)

  f32x4 accr[4][NT];
#pragma unroll
  for (int tl = 0; tl < 4; ++tl)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accr[tl][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (any) {
    const int nba = (total + GB - 1) / GB, nbd = (total_d + GB - 1) / GB;
    uint32_t wprev = 0x00ffffffu;  // the lower nodes (c + 1, one byte per axis) this thread's record of the previous batch had; 0xff = none
    const int rec_id = tid & (GB - 1);  // this thread's record of every batch; the two threads of a record split its channels
    const int half = tid >= GB;

    // -- the thread's part of the record pass: the record's column of the weight table and the set of tiles it touches (first
    // thread of the record), and its row of per-channel values.  A full-width record of an SH grid arrives COMPACT -- d density,
    // d raw r, g, b and the unit viewing direction -- and is expanded here, d raw[colour] * Y_k(v) in the operation order of the
    // reference's evaluate_spherical_harmonics, so that the expanded values never exist in HBM.
    auto stage = [&](auto expand_tag, float4 q0, float4 q1, float4 q2, int nrec) {
      constexpr bool EXPAND = decltype(expand_tag)::value;
      constexpr int RW = gather_record_words(EXPAND ? C4 : 4);
      if (half == 0) {
        uint32_t tm = 0, wnow = 0x00ffffffu;
        float* wcol = wtab + rec_id;
#pragma unroll
        for (int ax = 0; ax < 3; ++ax) {  // un-write the previous batch's entries (same thread, same column)
          const int c1 = (int)((wprev >> (8 * ax)) & 0xffu);  // c + 1
          if (c1 != 0xff) {
            if (c1 >= 1) wcol[(ax * 8 + c1 - 1) * GROW] = 0.0f;
            if (c1 < (BX == 4 && ax == 0 ? 4 : B)) wcol[(ax * 8 + c1) * GROW] = 0.0f;
          }
        }
        if (rec_id < nrec) {
          const float idx[3] = {q0.x, q0.y, q0.z};
          const int org[3] = {X0, Y0, Z0};
          uint32_t pb[3];
          wnow = 0;
#pragma unroll
          for (int ax = 0; ax < 3; ++ax) {
            const float fl = floorf(idx[ax]);
            const int c = (int)fl - org[ax];  // lower node of the cell relative to the brick: -1 .. B - 1
            const int sh = ax == 2 ? 2 : 1;
            uint32_t bits = 0;
            if (c >= 0) {
              wcol[(ax * 8 + c) * GROW] = (fl + 1.0f) - idx[ax];  // same arithmetic as locate()
              bits |= 1u << (c >> sh);
            }
            if (c + 1 < (BX == 4 && ax == 0 ? 4 : B)) {
              wcol[(ax * 8 + c + 1) * GROW] = idx[ax] - fl;
              bits |= 1u << ((c + 1) >> sh);
            }
            pb[ax] = bits;
            wnow |= (uint32_t)(c + 1) << (8 * ax);
          }
          uint32_t yz = 0;
#pragma unroll
          for (int py = 0; py < 4; ++py)
            if ((pb[1] >> py) & 1u) yz |= pb[2] << (py * npz);
#pragma unroll
          for (int px = 0; px < 4; ++px)
            if ((pb[0] >> px) & 1u) tm |= yz << (px * npy * npz);
        }
        wprev = wnow;
        s_tmask[rec_id] = tm;
      }
      float* row = rows + rec_id * RW;
      if constexpr (EXPAND) {
        // the first thread of the record (which also did the weights and the tile set) writes the base quad -- density and the
        // degree-0 channels, whose basis value is a constant --, the second evaluates the SH basis and writes the rest channels
        const float graw[3] = {q1.x, q1.y, q1.z};
        constexpr int NQ = C4 / 4;
        if (half == 0) {
          *reinterpret_cast<float4*>(row) = make_float4(q0.w, graw[0] * kC0, graw[1] * kC0, graw[2] * kC0);
        } else {
          float Y[16];
          sh_basis<K>(q1.w, q2.x, q2.y, Y);
#pragma unroll
          for (int q = 1; q < NQ; ++q) {
            float v[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] = (4 * q + x < C) ? record_channel<K>(4 * q + x, q0.w, graw, Y) : 0.0f;
            *reinterpret_cast<float4*>(row + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      } else {
        if (half == 0) *reinterpret_cast<float4*>(row) = q1;
      }
    };
    // -- bin the batch to the tiles it touches and multiply it into the tiles' accumulators (WIDE: rows of C4 channels, else 4)
    auto process = [&](auto wide_tag, int nrec) {
      constexpr bool WIDE = decltype(wide_tag)::value;
      constexpr int RW = gather_record_words(WIDE ? C4 : 4);  // words per row in LDS
      constexpr int NTW = WIDE ? NT : 1;
      RF_PROF_MARK(7);  // record pass (thread 0's own work)
      __syncthreads();
      RF_PROF_MARK(2);  // ... and its barrier
      // every wave lists the records of ITS tiles (ballot + prefix count: list order = record order) ...
      int cnt[4] = {0, 0, 0, 0};
#if defined(RF_BRICK_PROFILE) || defined(RF_BRICK_ABLATE)
      const bool no_lists = a.stagger & 0x200000, no_tiles = a.stagger & 0x100000;
#else
      constexpr bool no_lists = false, no_tiles = false;
#endif
      if (!no_lists)
#pragma unroll
      for (int wd = 0; wd < NW; ++wd) {
        if (wd * 64 >= nrec) break;  // (wave-uniform: the short batches at the end of a list fill one or two words only)
        const uint32_t tmv = s_tmask[wd * 64 + lane] >> wave;
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
          const bool hit = (tmv >> (NWV * tl)) & 1u;
          const unsigned long long m = __ballot(hit);
          const int pos = cnt[tl] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
          if (hit) s_list[(tl * NWV + wave)][pos] = (unsigned char)(wd * 64 + lane);
          cnt[tl] += __popcll(m);
        }
      }
#pragma unroll
      for (int tl = 0; tl < 4; ++tl)
        if (lane < 8) s_list[(tl * NWV + wave)][cnt[tl] + lane] = 0;  // pad the last instructions with a record that exists
      // ... and multiplies them into the tiles' accumulators, four records per instruction, two instructions' operands in flight
#pragma unroll
      for (int tl = 0; tl < 4; ++tl) {
        const int t = (tl * NWV + wave);
        const int n = cnt[tl];
        if (t < ntiles && n > 0 && !no_tiles) {
          const int pz = t % npz, py = (t / npz) % npy, px = t / (npz * npy);
          // this lane's three entries of a record's weight-table column, as byte offsets from the record's own
          const float* wx = wtab + (2 * px + mdx) * GROW;
          const float* wy = wtab + (8 + 2 * py + mdy) * GROW;
          const float* wz = wtab + (16 + 4 * pz + mdz) * GROW;
          const float* gbase = rows + (WIDE ? jj : (jj & 3));  // (base-channel rows: lanes 4..15 re-read channels 0..3)
          // software pipeline: the record index of instruction q + 2 and the operands of instruction q + 1 are requested before
          // instruction q is issued; an iteration waits once, at its top, for requests that are a whole iteration old
          const uint32_t a_list = lds_offset(s_list[t]) + kk;
          const uint32_t a_x = lds_offset(wx), a_y = lds_offset(wy), a_z = lds_offset(wz), a_g = lds_offset(gbase);
          const int nq = (n + 3) >> 2;
          // two register sets (A: even instructions, B: odd ones), no rotation copies.  WIDE: channels >= C4 of the second block
          // read into the row's padding / the next row; those accumulator columns are never stored
          uint32_t ra = 0, rb = 0;
          float aw0 = 0.f, aw1 = 0.f, aw2 = 0.f, bw0 = 0.f, bw1 = 0.f, bw2 = 0.f;
          float ag0 = 0.f, ag1 = 0.f, ag2 = 0.f, ag3 = 0.f, bg0 = 0.f, bg1 = 0.f, bg2 = 0.f, bg3 = 0.f;  // (blocks 2, 3: SH degree 3 only)
          auto request_ops = [&](float& w0, float& w1, float& w2, float& g0, float& g1, float& g2, float& g3, uint32_t r) {
            lds_request_f32<0>(w0, a_x + 4 * r);
            lds_request_f32<0>(w1, a_y + 4 * r);
            lds_request_f32<0>(w2, a_z + 4 * r);
            const uint32_t a_row = a_g + (uint32_t)__umul24(r, RW * 4);  // (r is a byte: a full-rate 24-bit multiply-add, not a 64-bit one)
            lds_request_f32<0>(g0, a_row);
            if constexpr (NTW > 1) lds_request_f32<64>(g1, a_row);
            if constexpr (NTW > 2) {
              lds_request_f32<128>(g2, a_row);
              lds_request_f32<192>(g3, a_row);
            }
          };
          auto wait_ops = [&](float& w0, float& w1, float& w2, float& g0, float& g1, float& g2, float& g3, uint32_t& r) {
            if constexpr (NTW > 2)
              lds_wait(w0, w1, w2, g0, g1, g2, g3, r);
            else
              lds_wait(w0, w1, w2, g0, g1, r);
          };
          auto multiply = [&](float w0, float w1, float w2, float g0, float g1, float g2, float g3, int q) {
            const float wv = (w0 * w1) * w2;
            const float wa = (4 * q + kk < n) ? wv : 0.0f;
            const float ga = (WIDE || jj < 4) ? g0 : 0.0f;
            accr[tl][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, ga, accr[tl][0], 0, 0, 0);
            if constexpr (NTW > 1) accr[tl][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, g1, accr[tl][1], 0, 0, 0);
            if constexpr (NTW > 2) {
              accr[tl][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, g2, accr[tl][2], 0, 0, 0);
              accr[tl][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, g3, accr[tl][3], 0, 0, 0);
            }
          };
          lds_request_u8(ra, a_list);      // record index of instruction 0
          lds_request_u8(rb, a_list + 4);  // ... 1
          lds_wait(ra, rb);
          request_ops(aw0, aw1, aw2, ag0, ag1, ag2, ag3, ra);
          for (int q = 0; q < nq; q += 2) {
            wait_ops(aw0, aw1, aw2, ag0, ag1, ag2, ag3, rb);
            lds_request_u8(ra, a_list + 4 * q + 8);  // index of instruction q + 2 (the list has slack behind its capacity)
            request_ops(bw0, bw1, bw2, bg0, bg1, bg2, bg3, rb);
            multiply(aw0, aw1, aw2, ag0, ag1, ag2, ag3, q);
            if (q + 1 < nq) {
              wait_ops(bw0, bw1, bw2, bg0, bg1, bg2, bg3, ra);
              lds_request_u8(rb, a_list + 4 * q + 12);
              request_ops(aw0, aw1, aw2, ag0, ag1, ag2, ag3, ra);
              multiply(bw0, bw1, bw2, bg0, bg1, bg2, bg3, q + 1);
            }
          }
          wait_ops(aw0, aw1, aw2, ag0, ag1, ag2, ag3, ra);  // nothing of this tile stays in flight
          wait_ops(bw0, bw1, bw2, bg0, bg1, bg2, bg3, rb);
        }
      }
      RF_PROF_MARK(3);  // lists + tiles (MFMA), thread 0's own work
      __syncthreads();
      RF_PROF_MARK(2);
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;

    // global loads of a batch straight into registers -- every thread the quads of ITS record (the two threads of a record
    // both: the second copy is an L1 hit) --, issued one batch ahead of their use; always unconditional and clamped, with their
    // own registers per list kind (a conditionally assigned register is merged with a copy, and the copy waits for the load)
    int sri = 0, dri = 0;  // running range index of this thread (its records only move forward), the range's bounds cached
    int rlo = 0, rhi = 0, dlo = 0, dhi = 0;
    const float4* rptr = a.wide[SPLIT ? min(part, kMaxListsPerKind - 1) : 0].rec;    // list base + (start of the range - its position in the concatenation)
    const float4* dptr = a.narrow[SPLIT ? min(part, kMaxListsPerKind - 1) : 0].rec;
    float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0, w2 = w0, d0 = w0, d1 = w0;
    // the range that holds record v of the concatenation: the largest i with cum[i] <= v.  One list (15 ranges, the single-GPU case):
    // the whole table in four independent 16-byte reads and 15 compares -- the walk below is a chain of dependent LDS reads
    auto range_of = [&](const int* cum, int v, int nlists, int from) -> int {
      if (nlists == 1) {  // (wave-uniform)
        const int4* c4 = reinterpret_cast<const int4*>(cum);
        const int4 c0 = c4[0], c1 = c4[1], c2 = c4[2], c3 = c4[3];
        return (c0.y <= v) + (c0.z <= v) + (c0.w <= v) + (c1.x <= v) + (c1.y <= v) + (c1.z <= v) + (c1.w <= v) + (c2.x <= v) + (c2.y <= v) + (c2.z <= v) +
               (c2.w <= v) + (c3.x <= v) + (c3.y <= v) + (c3.z <= v);  // (cum[15] = the total > v)
      }
      int i = from;
      while (cum[i] > v) --i;  // (only when the last batch is fetched a second time)
      while (cum[i + 1] <= v) ++i;
      return i;
    };
    auto fetch_wide = [&](int s) {
      const int nrec = min(GB, total - s * GB);
      const int v = s * GB + min(rec_id, nrec - 1);  // record of the concatenated ranges
      if (v < rlo || v >= rhi) {  // the range of the previous batch's record no longer holds this one
        sri = range_of(s_wcum, v, nwide_p, sri);
        rlo = s_wcum[sri];
        rhi = s_wcum[sri + 1];
        rptr = ((!SPLIT && a.nwide == 1) ? a.wide[0].rec : a.wide[list_index(sri / kRangeEntries)].rec) + (long long)(s_wstart[sri] - rlo) * QW;
      }
      const float4* p = rptr + (long long)v * QW;
      w0 = load_f4<RF_NT_RECORD_LOAD>(p);
      w1 = load_f4<RF_NT_RECORD_LOAD>(p + 1);
      if constexpr (QW > 2) w2 = load_f4<RF_NT_RECORD_LOAD>(p + 2);
    };
    auto fetch_narrow = [&](int sd) {
      const int nrec = min(GB, total_d - sd * GB);
      const int v = sd * GB + min(rec_id, nrec - 1);
      if (v < dlo || v >= dhi) {
        dri = range_of(s_ncum, v, nnarrow_p, dri);
        dlo = s_ncum[dri];
        dhi = s_ncum[dri + 1];
        dptr = ((!SPLIT && a.nnarrow == 1) ? a.narrow[0].rec : a.narrow[list_index(dri / kRangeEntries)].rec) + (long long)(s_nstart[dri] - dlo) * QN;
      }
      const float4* p = dptr + (long long)v * QN;
      d0 = load_f4<RF_NT_RECORD_LOAD>(p);
      d1 = load_f4<RF_NT_RECORD_LOAD>(p + 1);
    };
    if (nbd > 0) fetch_narrow(0);
    if (nba > 0) fetch_wide(0);
    // (behind the first batches' loads:) the weight table starts zero-filled; a record's thread clears the entries of the previous
    // batch before it writes new ones
    for (int i = tid; i < GROW * kGatherWtab / 4; i += TH) reinterpret_cast<float4*>(wtab)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    for (int s = 0; s < nba; ++s) {
      const int nrec = min(GB, total - s * GB);
      if constexpr (K > 1)
        stage(Yes{}, w0, w1, w2, nrec);
      else
        stage(No{}, w0, w1, w2, nrec);
      RF_PROF_MARK(1);  // waiting for the batch's loads, record pass
      fetch_wide(min(s + 1, nba - 1));  // (the last batch again at the end: cheaper than a conditional)
      RF_PROF_MARK(6);  // issuing the next batch's loads
      process(Yes{}, nrec);
    }
    for (int sd = 0; sd < nbd; ++sd) {
      const int nrec = min(GB, total_d - sd * GB);
      stage(No{}, d0, d1, d1, nrec);
      RF_PROF_MARK(1);
      fetch_narrow(min(sd + 1, nbd - 1));
      RF_PROF_MARK(6);
      process(No{}, nrec);
    }
    // -- the accumulator image for the flush (the batch buffers are dead).  Accumulator register e of a lane: node row
    // 4 (lane >> 4) + e of the tile, channel 16 nt + (lane & 15)
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
      const int t = (tl * NWV + wave);
      if (t < ntiles) {
        const int pz = t % npz, py = (t / npz) % npy, px = t / (npz * npy);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int ch = 16 * nt + jj;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * kk + e;
            const int nx = 2 * px + (i >> 3), ny = 2 * py + ((i >> 2) & 1), nz = 4 * pz + (i & 3);
            if (ch < CS) acc[nx * SX + ny * SY + nz * CS + ch] = accr[tl][nt][e];
          }
        }
      }
    }
    __syncthreads();
    RF_PROF_MARK(5);  // accumulator image
  }
  if constexpr (SPLIT) {
    // The parts of a brick meet here.  Every part that summed something writes its image to global memory; the last one to arrive
    // (arrival counter) adds the others' images to its own and flushes the brick.  Nobody waits for anybody: no spinning, no
    // assumption about the dispatch order.  The images and flags cross workgroups -- and XCDs, each with an L2 of its own -- inside
    // ONE launch: they are written and read with device-scope relaxed atomics (sc1 accesses: written through / read past the local
    // L2) and ordered by vmcnt(0) + the workgroup barrier in front of the counter's atomic.  (A device-scope FENCE instead -- the
    // textbook last-block pattern -- writes back and invalidates the whole L2 of the XCD per workgroup: measured 0.6 us per
    // workgroup, serialised: 1.1 ms for 2048 workgroups.)
    const int words = brick_acc_words(B, C);
    int* state = a.part_state + (long long)brick_local * (1 + parts);
    unsigned long long* images = reinterpret_cast<unsigned long long*>(a.partial + (long long)brick_local * parts * words);
    const int pairs = words / 2;
    __shared__ int s_arrival;
    if (any) {
      unsigned long long* mine = images + (long long)part * pairs;
      const unsigned long long* src = reinterpret_cast<const unsigned long long*>(acc);
      for (int i = tid; i < pairs; i += TH) __hip_atomic_store(mine + i, src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (tid == 0) __hip_atomic_store(&state[1 + part], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's stores have been performed at device scope
    __syncthreads();
    if (tid == 0) s_arrival = __hip_atomic_fetch_add(&state[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_arrival != parts - 1) return;  // (uniform for the whole workgroup)
    for (int j = 0; j < parts; ++j) {
      if (j == part) continue;
      if (__hip_atomic_load(&state[1 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) continue;  // (uniform)
      const unsigned long long* other = images + (long long)j * pairs;
      float2* own = reinterpret_cast<float2*>(acc);
      for (int i = tid; i < pairs; i += TH) {
        const unsigned long long o = __hip_atomic_load(other + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float2 v = any ? own[i] : make_float2(0.f, 0.f);
        v.x += __uint_as_float((uint32_t)o);
        v.y += __uint_as_float((uint32_t)(o >> 32));
        own[i] = v;
      }
      any = true;
      __syncthreads();
    }
    if (tid <= parts) state[tid] = 0;  // left clean for the next launch
    __syncthreads();
  }
#if defined(RF_BRICK_PROFILE) || defined(RF_BRICK_ABLATE)
  if (!(a.stagger & 0x400000))  // (ablation: the batch phases alone)
#endif
  brick_flush<K, ADAM, ONE_ROUND, TH, BX, MIRROR>(g, a, acc, any, X0, Y0, Z0, gdens, gfeat);
  RF_PROF_MARK(4);  // flush (stores issued, not necessarily retired)
  RF_PROF_END();
}

// =============================================================================================
// stage transition: scale_voxel_grid_with_required_output_size (thre3d_reprs/voxels.py:334-373) = F.interpolate(mode="trilinear",
// align_corners=False) of the [F+1]-channel volume.  One thread per (destination node, channel), channel fastest; source and
// destination in any storage (the value lands in the destination's layout directly: no unified tensor, no permutes).  ATen's
// arithmetic (UpSampleKernel.cpp / UpSample.h): scale = in / out in float, src = fma(scale, dst + 0.5, -0.5) clamped at 0,
// i0 = min(floor(src), in - 1), i1 = i0 + (i0 < in - 1), lambda = clamp(src - i0, 0, 1); the 8 corners weighted by the products of
// the per-axis weights and summed in the order of ATen's channels-last CPU kernel (see below).
// =============================================================================================
__device__ __forceinline__ long long channel_offset(const GridArgs& g, unsigned int lin, int ch, int K, bool& in_first) {
  // ch: 0 = density, 1..3 = degree-0 coefficient of r, g, b, then colour-major higher degrees (the order of a node's accumulators)
  if (g.layout == RF_LAYOUT_REFERENCE) {
    in_first = ch == 0;
    if (ch == 0) return (long long)lin * g.dstride;
    if (ch < 4) return (long long)lin * g.fstride + (long long)(ch - 1) * K;
    const int j = ch - 4, colour = j / (K - 1), kk = 1 + j - colour * (K - 1);
    return (long long)lin * g.fstride + (long long)colour * K + kk;
  }
  in_first = ch < 4;
  return ch < 4 ? (long long)lin * g.dstride + ch : (long long)lin * g.fstride + (ch - 4);
}

struct UpsampleAxis {
  int i0, i1;
  float w0, w1;
};
__device__ __forceinline__ UpsampleAxis upsample_axis(int dst, int in_size, float scale) {
  float src = fmaf(scale, (float)dst + 0.5f, -0.5f);  // (contracted in ATen's build)
  src = src < 0.0f ? 0.0f : src;
  UpsampleAxis a;
  a.i0 = min((int)floorf(src), in_size - 1);
  a.i1 = a.i0 + (a.i0 < in_size - 1 ? 1 : 0);
  const float lambda = fminf(fmaxf(src - (float)a.i0, 0.0f), 1.0f);
  a.w0 = 1.0f - lambda;
  a.w1 = lambda;
  return a;
}

__global__ void upsample_grid_kernel(GridArgs src, GridArgs dst, float* dst_first, float* dst_second, long long total) {
  const int C = dst.F + 1, K = dst.F / 3;
  const float sx = (float)src.X / (float)dst.X, sy = (float)src.Y / (float)dst.Y, sz = (float)src.Z / (float)dst.Z;
  for (long long it = (long long)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x * blockDim.x) {
    const int ch = (int)(it % C);
    long long node = it / C;
    const int z = (int)(node % dst.Z);
    node /= dst.Z;
    const int y = (int)(node % dst.Y), x = (int)(node / dst.Y);
    const UpsampleAxis ax = upsample_axis(x, src.X, sx), ay = upsample_axis(y, src.Y, sy), az = upsample_axis(z, src.Z, sz);
    auto at = [&](int xi, int yi, int zi) -> float {
      bool first;
      const long long off = channel_offset(src, node_lin(src, xi, yi, zi), ch, K, first);
      return (first ? src.dens : src.feat)[off];
    };
    // corner k = (dx, dy, dz) with z fastest, weight = (wx * wy) * wz
    float val[8], wgt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int dx = k >> 2, dy = (k >> 1) & 1, dz = k & 1;
      val[k] = at(dx ? ax.i1 : ax.i0, dy ? ay.i1 : ay.i0, dz ? az.i1 : az.i0);
      wgt[k] = ((dx ? ax.w1 : ax.w0) * (dy ? ay.w1 : ay.w0)) * (dz ? az.w1 : az.w0);
    }
    // The reference hands F.interpolate a channels-last volume (features in the reference order, density last), and ATen's
    // channels-last CPU kernel sums the 8 weighted corners right to left in its 8-wide vector body and left to right in the
    // scalar tail of the last C mod 8 channels, every step contracted to a fused multiply-add.  Both orders are reproduced per
    // channel, so that the result equals the reference's own output bit for bit (golden G3).
    int uc;  // index of this channel in the reference's unified [features, density] order
    if (ch == 0) {
      uc = dst.F;
    } else if (ch < 4) {
      uc = (ch - 1) * K;
    } else {
      const int j = ch - 4, colour = j / (K - 1);
      uc = colour * K + 1 + (j - colour * (K - 1));
    }
    float v;
    if (uc < (C & ~7)) {
      v = fmaf(val[7], wgt[7], val[6] * wgt[6]);
#pragma unroll
      for (int k = 5; k >= 0; --k) v = fmaf(val[k], wgt[k], v);
    } else {
      v = fmaf(val[0], wgt[0], val[1] * wgt[1]);
#pragma unroll
      for (int k = 2; k < 8; ++k) v = fmaf(val[k], wgt[k], v);
    }
    bool first;
    const long long off = channel_offset(dst, node_lin(dst, x, y, z), ch, K, first);
    (first ? dst_first : dst_second)[off] = v;
  }
}

// Re-layout of a whole grid: every (node, channel) of `src` copied to where `dst`'s storage keeps it (same dims, same F).  One
// thread per element, channel fastest.  Used to keep a split-layout SHADOW of a grid held in the reference's two tensors: the
// forward passes gather from the shadow (aligned 16-byte base records, density and degree-0 colour in one gather), the
// gradients are produced in the layout of the Parameters.
__global__ void convert_grid_kernel(GridArgs src, GridArgs dst, float* dst_first, float* dst_second, unsigned int nodes) {
  // threadIdx.x = channel (32 or 64 lanes per node, the ones beyond F + 1 idle), threadIdx.y = node within the block: no
  // division for the channel, 32-bit index arithmetic throughout (node counts are < 2^32)
  const int C = dst.F + 1, K = dst.F / 3;
  const int ch = threadIdx.x;
  const bool linear = !src.bricked && !dst.bricked;  // node index = linear index in both
  for (unsigned int node = blockIdx.x * blockDim.y + threadIdx.y; node < nodes; node += gridDim.x * blockDim.y) {
    if (ch >= C) continue;
    unsigned int ls = node, ld = node;
    if (!linear) {
      const unsigned int z = node % (unsigned)dst.Z, t = node / (unsigned)dst.Z;
      const unsigned int y = t % (unsigned)dst.Y, x = t / (unsigned)dst.Y;
      ls = node_lin(src, (int)x, (int)y, (int)z);
      ld = node_lin(dst, (int)x, (int)y, (int)z);
    }
    bool sf, df;
    const long long so = channel_offset(src, ls, ch, K, sf);
    const long long doff = channel_offset(dst, ld, ch, K, df);
    (df ? dst_first : dst_second)[doff] = (sf ? src.dens : src.feat)[so];
  }
}

// The common case of the re-layout -- reference tensors -> split tensors, both in linear node order, whole quads per node (SH degree
// 0 or 2) -- one thread per destination float4: quad 0 of a node = (density, degree-0 r, g, b), quad q >= 1 = rest[4 (q - 1) ..].
// The four source floats are 4-byte gathers inside the node's 108-byte feature record (cache hits); the store is a coalesced 16 B.
template <int K>
__global__ void reference_to_split_kernel(const float* __restrict__ dens, const float* __restrict__ feat, long long dstride, long long fstride,
                                          float4* __restrict__ base, float4* __restrict__ rest, unsigned int nodes) {
  constexpr int QN = (3 * K + 1) / 4;  // quads per node
  constexpr int KR = K > 1 ? K - 1 : 1;
  const unsigned int total = nodes * QN;
  for (unsigned int it = blockIdx.x * blockDim.x + threadIdx.x; it < total; it += gridDim.x * blockDim.x) {
    const unsigned int node = it / QN, q = it - node * QN;
    const float* f = feat + (long long)node * fstride;
    if (q == 0) {
      base[node] = make_float4(dens[(long long)node * dstride], f[0], f[K], f[2 * K]);
    } else {
      float v[4];
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const unsigned int r = 4 * (q - 1) + x;
        const unsigned int colour = r / KR, k = r - colour * KR + 1;
        v[x] = f[colour * K + k];
      }
      rest[(long long)node * (QN - 1) + (q - 1)] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// ... and back: split tensors -> reference tensors, one thread per SOURCE float4 (coalesced 16-byte load, four 4-byte stores inside the
// node's feature record).  Used when the split shadow is the copy the optimizer updated (deferred gradients, optim.FlatGrid).
template <int K>
__global__ void split_to_reference_kernel(const float4* __restrict__ base, const float4* __restrict__ rest, float* __restrict__ dens,
                                          float* __restrict__ feat, long long dstride, long long fstride, unsigned int nodes) {
  constexpr int QN = (3 * K + 1) / 4;
  constexpr int KR = K > 1 ? K - 1 : 1;
  const unsigned int total = nodes * QN;
  for (unsigned int it = blockIdx.x * blockDim.x + threadIdx.x; it < total; it += gridDim.x * blockDim.x) {
    const unsigned int node = it / QN, q = it - node * QN;
    float* f = feat + (long long)node * fstride;
    if (q == 0) {
      const float4 v = base[node];
      dens[(long long)node * dstride] = v.x;
      f[0] = v.y;
      f[K] = v.z;
      f[2 * K] = v.w;
    } else {
      const float4 v4 = rest[(long long)node * (QN - 1) + (q - 1)];
      const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        const unsigned int r = 4 * (q - 1) + x;
        const unsigned int colour = r / KR, k = r - colour * KR + 1;
        f[colour * K + k] = v[x];
      }
    }
  }
}

// The same re-layout with coalesced 16-byte STORES for contiguous reference tensors (features [nodes * F] and densities [nodes] written
// as float4s, each gathering its four source floats -- cache hits inside the nodes' base / rest records): 0.164 -> ms of the per-element
// store version on the 128^3 / SH-2 grid.
template <int K>
__global__ void split_to_reference_quads_kernel(const float* __restrict__ base, const float* __restrict__ rest, float4* __restrict__ dens4,
                                                float4* __restrict__ feat4, unsigned int nodes) {
  constexpr unsigned int F = 3 * K, R = F - 3;
  constexpr unsigned int KR = K > 1 ? K - 1 : 1;
  const unsigned int fq = nodes * F / 4, dq = nodes / 4;  // (host-checked: nodes % 4 == 0)
  for (unsigned int it = blockIdx.x * blockDim.x + threadIdx.x; it < fq + dq; it += gridDim.x * blockDim.x) {
    float v[4];
    if (it < fq) {
#pragma unroll
      for (unsigned int x = 0; x < 4; ++x) {
        const unsigned int f = 4 * it + x, node = f / F, c = f - node * F;
        const unsigned int colour = c / K, k = c - colour * K;
        v[x] = (k == 0) ? base[node * 4 + 1 + colour] : rest[node * R + colour * KR + (k - 1)];
      }
      feat4[it] = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      const unsigned int n0 = 4 * (it - fq);
      dens4[it - fq] = make_float4(base[n0 * 4], base[n0 * 4 + 4], base[n0 * 4 + 8], base[n0 * 4 + 12]);
    }
  }
}

// =============================================================================================
// standalone point query: VoxelGrid.forward (thre3d_reprs/voxels.py:276-331) and its adjoint.
// One thread per (point, output channel); channel c < F is feature c in the reference order (colour*K + k),
// channel F is the activated density.  Any point is allowed (zeros padding outside the grid, no AABB mask --
// the mask belongs to process_points).  Not a hot path: the renderer uses the fused kernels above.
// =============================================================================================
struct QueryCorner {
  long long lin;
  float w;
  bool ok;
};

__device__ __forceinline__ void query_cell(const float p[3], const GridArgs& g, int i0[3], float w0[3], float w1[3]) {
  const int dims[3] = {g.X, g.Y, g.Z};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float q = p[a] * g.nscale[a] + g.nbias[a];
    const float idx = ((q + 1.0f) * (float)dims[a] - 1.0f) / 2.0f;
    const float fl = floorf(idx);
    w1[a] = idx - fl;
    w0[a] = (fl + 1.0f) - idx;
    // far-away points: every corner is outside anyway; clamp so that the integer conversion is defined
    i0[a] = (int)fminf(fmaxf(fl, -2.0f), (float)dims[a]);
  }
}

__device__ __forceinline__ QueryCorner query_corner(int k, const int i0[3], const float w0[3], const float w1[3],
                                                    const GridArgs& g) {
  const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
  const int ix = i0[0] + dx, iy = i0[1] + dy, iz = i0[2] + dz;
  QueryCorner c;
  c.ok = ix >= 0 && ix < g.X && iy >= 0 && iy < g.Y && iz >= 0 && iz < g.Z;
  c.lin = node_lin(g, min(max(ix, 0), g.X - 1), min(max(iy, 0), g.Y - 1), min(max(iz, 0), g.Z - 1));
  c.w = ((dx ? w1[0] : w0[0]) * (dy ? w1[1] : w0[1])) * (dz ? w1[2] : w0[2]);
  return c;
}

// element offset of output channel c of voxel `lin` inside the tensor that holds it; *in_first = it is the
// densities/base tensor rather than the features/rest tensor
__device__ __forceinline__ long long query_offset(int c, long long lin, const GridArgs& g, bool* in_first) {
  const int F = g.F, K = F / 3;
  if (c == F) {
    *in_first = true;
    return lin * g.dstride;
  }
  if (g.layout == RF_LAYOUT_SPLIT) {
    const int ch = c / K, k = c % K;
    if (k == 0) {
      *in_first = true;
      return lin * g.dstride + 1 + ch;
    }
    *in_first = false;
    return lin * g.fstride + ch * (K - 1) + (k - 1);
  }
  *in_first = false;
  return lin * g.fstride + c;
}

__device__ __forceinline__ float density_post(float pre, int mode) {
  if (mode == RF_DENSITY_RELU) return fmaxf(pre, 0.0f);
  if (mode == RF_DENSITY_SOFTPLUS) return (pre > 20.0f) ? pre : log1pf(expf(pre));
  return pre;
}

__global__ void grid_query_kernel(GridArgs g, const float* __restrict__ points, long long n, float* __restrict__ out) {
  const int C = g.F + 1;
  const long long total = n * C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long pt = idx / C;
    const int c = (int)(idx % C);
    const float p[3] = {points[pt * 3], points[pt * 3 + 1], points[pt * 3 + 2]};
    int i0[3];
    float w0[3], w1[3];
    query_cell(p, g, i0, w0, w1);
    const bool dens = (c == g.F);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const QueryCorner qc = query_corner(k, i0, w0, w1, g);
      if (qc.ok) {
        bool first;
        const long long off = query_offset(c, qc.lin, g, &first);
        float v = first ? g.dens[off] : g.feat[off];
        if (dens) {
          v = v * g.rho;
          if (g.mode == RF_DENSITY_ABS) v = fabsf(v);
        }
        acc = acc + v * qc.w;
      }
    }
    out[idx] = dens ? density_post(acc, g.mode) : acc;
  }
}

__global__ void grid_query_backward_kernel(GridArgs g, const float* __restrict__ points, long long n,
                                           const float* __restrict__ gout, float* gfirst, float* gsecond) {
  const int C = g.F + 1;
  const long long total = n * C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long pt = idx / C;
    const int c = (int)(idx % C);
    float go = gout[idx];
    if (go == 0.0f) continue;
    const float p[3] = {points[pt * 3], points[pt * 3 + 1], points[pt * 3 + 2]};
    int i0[3];
    float w0[3], w1[3];
    query_cell(p, g, i0, w0, w1);
    const bool dens = (c == g.F);
    if (dens && (g.mode == RF_DENSITY_RELU || g.mode == RF_DENSITY_SOFTPLUS)) {
      float pre = 0.0f;  // the activation derivative needs the interpolated pre-activation
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const QueryCorner qc = query_corner(k, i0, w0, w1, g);
        if (qc.ok) pre = pre + (g.dens[qc.lin * g.dstride] * g.rho) * qc.w;
      }
      if (g.mode == RF_DENSITY_RELU)
        go = (pre > 0.0f) ? go : 0.0f;
      else
        go = go * ((pre > 20.0f) ? 1.0f : 1.0f / (1.0f + expf(-pre)));
      if (go == 0.0f) continue;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const QueryCorner qc = query_corner(k, i0, w0, w1, g);
      if (!qc.ok) continue;
      bool first;
      const long long off = query_offset(c, qc.lin, g, &first);
      float gv = qc.w * go;
      if (dens) {
        gv = gv * g.rho;
        if (g.mode == RF_DENSITY_ABS) {
          const float dv = g.dens[off] * g.rho;
          gv = (dv > 0.f) ? gv : ((dv < 0.f) ? -gv : 0.0f);
        }
      }
      if (gv != 0.0f) unsafeAtomicAdd((first ? gfirst : gsecond) + off, gv);
    }
  }
}

// =============================================================================================
// ray generation (rendering/volumetric/utils/misc.py:12-50)
// =============================================================================================
struct Pose {
  float r[9];
  float t[3];
};


__global__ void cast_rays_kernel(int H, int W, float focal, Pose pose, float* origins, float* dirs) {
  const long long n = (long long)H * W;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(p / W), j = (int)(p % W);
    float d[3];
    pixel_ray(i, j, H, W, focal, pose.r, d);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      origins[p * 3 + a] = pose.t[a];
      dirs[p * 3 + a] = d[a];
    }
  }
}

__global__ void cast_selected_rays_kernel(int H, int W, float focal, const float* poses, int num_poses,
                                          const int64_t* pix, long long n, float* origins, float* dirs) {
  const long long hw = (long long)H * W;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
    const long long p = pix[q];
    int b = (int)(p / hw);
    b = min(max(b, 0), num_poses - 1);
    const long long rem = p - (long long)b * hw;
    const int i = (int)(rem / W), j = (int)(rem % W);
    const float* P = poses + b * 12;  // [3, 4] = rotation | translation
    const float R[9] = {P[0], P[1], P[2], P[4], P[5], P[6], P[8], P[9], P[10]};
    float d[3];
    pixel_ray(i, j, H, W, focal, R, d);
    origins[q * 3 + 0] = P[3];
    origins[q * 3 + 1] = P[7];
    origins[q * 3 + 2] = P[11];
#pragma unroll
    for (int a = 0; a < 3; ++a) dirs[q * 3 + a] = d[a];
  }
}

// ---------------------------------------------------------------------------------------------
// fused batch selection: the r-th ray of the batch is pixel PRP_key(r) of the B*H*W pixels of the image batch,
// where PRP is a keyed bijection of [0, P) (4-round Feistel network on ceil(log2 P) bits + cycle walking).
// The first R values of a random permutation = R distinct uniformly random pixels: the same sampling law as
// torch.randperm(P)[:R] (utils/misc.py:117-129) without sorting P keys.
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ unsigned long long keyed_permutation(unsigned long long i, unsigned long long P, int bits,
                                                                unsigned long long key) {
  const int hr = bits / 2, hl = bits - hr;  // x = L (hl bits) | R (hr bits)
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  unsigned long long x = i;
  do {
    uint32_t L = (uint32_t)(x >> hr), R = (uint32_t)(x & ((1ull << hr) - 1ull));
    // two double-rounds; after each double-round the halves have their original widths again
#pragma unroll
    for (int rnd = 0; rnd < 2; ++rnd) {
      L ^= mix32(R ^ k0 ^ (0x9E3779B9u * (2 * rnd + 1))) & (uint32_t)((1ull << hl) - 1ull);
      R ^= mix32(L ^ k1 ^ (0x85EBCA6Bu * (2 * rnd + 2))) & (uint32_t)((1ull << hr) - 1ull);
    }
    x = ((unsigned long long)L << hr) | R;
  } while (x >= P);  // cycle walking keeps it a bijection of [0, P); < 2 trips on average
  return x;
}

__global__ void select_rays_and_pixels_kernel(int H, int W, float focal, const float* poses, const int64_t* image_ids,
                                              int num_batch_images, const float* pixel_table, unsigned long long key,
                                              int bits, long long first, long long n, float* origins, float* dirs,
                                              float* pixels, int64_t* pixel_index, float* zero4) {
  if (zero4 && blockIdx.x == 0 && threadIdx.x < 4) zero4[threadIdx.x] = 0.0f;  // (rf_train_step: the loss sums of the iteration)
  const long long hw = (long long)H * W;
  const unsigned long long P = (unsigned long long)num_batch_images * hw;
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
    const long long p = (long long)keyed_permutation((unsigned long long)(first + q), P, bits, key);
    const int b = (int)(p / hw);
    const long long rem = p - (long long)b * hw;
    const int i = (int)(rem / W), j = (int)(rem % W);
    const long long img = image_ids ? image_ids[b] : b;
    const float* Pm = poses + img * 12;  // [3, 4] = rotation | translation
    const float R[9] = {Pm[0], Pm[1], Pm[2], Pm[4], Pm[5], Pm[6], Pm[8], Pm[9], Pm[10]};
    float d[3];
    pixel_ray(i, j, H, W, focal, R, d);
    origins[q * 3 + 0] = Pm[3];
    origins[q * 3 + 1] = Pm[7];
    origins[q * 3 + 2] = Pm[11];
    const float* px = pixel_table + (img * hw + rem) * 3;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      dirs[q * 3 + a] = d[a];
      pixels[q * 3 + a] = px[a];
    }
    if (pixel_index) pixel_index[q] = p;
  }
}

struct Box {
  float lo[3], hi[3];
};

__global__ void ray_aabb_bounds_kernel(const float* origins, const float* dirs, long long n, float near, float far,
                                       Box box, float* bounds, float* hit) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (long long)gridDim.x * blockDim.x) {
    const float o[3] = {origins[q * 3], origins[q * 3 + 1], origins[q * 3 + 2]};
    const float d[3] = {dirs[q * 3], dirs[q * 3 + 1], dirs[q * 3 + 2]};
    float t0, t1;
    const bool h = ray_box(o, d, box.lo, box.hi, near, far, t0, t1);
    bounds[q * 2] = t0;
    bounds[q * 2 + 1] = t1;
    if (hit) hit[q] = h ? 1.0f : 0.0f;
  }
}

// =============================================================================================
// occupancy mask (exact empty-space skipping for the ReLU field)
// =============================================================================================
__global__ void build_occupancy_kernel(GridArgs g, float threshold, uint32_t* occ, long long nwords) {
  // one lane per cell, 64 consecutive cells (z fastest) per wave -> two mask words per __ballot; the 8 nodes of
  // neighbouring cells overlap, so the loads hit L1
  const long long ncell = (long long)(g.X + 1) * (g.Y + 1) * (g.Z + 1);
  const long long nwave_items = (ncell + 63) / 64;
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long wave_stride = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long item = wave0; item < nwave_items; item += wave_stride) {
    const long long cell = item * 64 + lane;
    bool occ_cell = false;
    if (cell < ncell) {
      occ_cell = (g.mode != RF_DENSITY_RELU);
      if (!occ_cell) {
        const int cz = (int)(cell % (g.Z + 1));
        const long long t = cell / (g.Z + 1);
        const int cy = (int)(t % (g.Y + 1)), cx = (int)(t / (g.Y + 1));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int x = cx - 1 + (k & 1), y = cy - 1 + ((k >> 1) & 1), z = cz - 1 + (k >> 2);
          if (x >= 0 && x < g.X && y >= 0 && y < g.Y && z >= 0 && z < g.Z)
            occ_cell = occ_cell || (g.dens[(long long)node_lin(g, x, y, z) * g.dstride] * g.rho > threshold);
        }
      }
    }
    const unsigned long long bits = __ballot(occ_cell);
    if (lane == 0 && item * 2 < nwords) occ[item * 2] = (uint32_t)bits;
    if (lane == 32 && item * 2 + 1 < nwords) occ[item * 2 + 1] = (uint32_t)(bits >> 32);
  }
}

// =============================================================================================
// fused Adam (torch.optim.Adam, no weight decay, no amsgrad)
// =============================================================================================
// STREAM = true (all four pointers 16-byte aligned): one float4 per thread; the gradient and the two moments are
// streamed with non-temporal loads/stores (nothing reads them again before 700 MB of other traffic has passed), the
// parameters with ordinary ones (the next forward pass reads them).  Measured on MI355X (tools/adam_microbench.hip):
// 6.2-6.4 TB/s vs 5.8-6.0 TB/s for the plain grid-stride loop.
template <bool ZERO_GRAD, bool STREAM>
__global__ void adam_kernel(float* __restrict__ p, float* __restrict__ gradp, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float bc1,
                            float inv_bc2_sqrt) {
  const long long n4 = n / 4;
  const float step = lr / bc1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f4u gg, pp, mm, vv;
    if (STREAM) {
      const vf4 g4 = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(gradp) + i);
      const vf4 p4 = reinterpret_cast<const vf4*>(p)[i];
      const vf4 m4 = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(m) + i);
      const vf4 v4 = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(v) + i);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        gg.v[c] = g4[c];
        pp.v[c] = p4[c];
        mm.v[c] = m4[c];
        vv.v[c] = v4[c];
      }
    } else {
      // 16-byte accesses that only assume 4-byte alignment (a slice of the flat buffer may start anywhere)
      gg = reinterpret_cast<const f4u*>(gradp)[i];
      pp = reinterpret_cast<f4u*>(p)[i];
      mm = reinterpret_cast<f4u*>(m)[i];
      vv = reinterpret_cast<f4u*>(v)[i];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      adam_update(pp.v[c], mm.v[c], vv.v[c], gg.v[c], step, b1, b2, eps, inv_bc2_sqrt);
    }
    if (STREAM) {
      const vf4 p4 = {pp.v[0], pp.v[1], pp.v[2], pp.v[3]}, m4 = {mm.v[0], mm.v[1], mm.v[2], mm.v[3]}, v4 = {vv.v[0], vv.v[1], vv.v[2], vv.v[3]};
      reinterpret_cast<vf4*>(p)[i] = p4;
      __builtin_nontemporal_store(m4, reinterpret_cast<vf4*>(m) + i);
      __builtin_nontemporal_store(v4, reinterpret_cast<vf4*>(v) + i);
      if (ZERO_GRAD) {
        const vf4 z4 = {0.f, 0.f, 0.f, 0.f};
        __builtin_nontemporal_store(z4, reinterpret_cast<vf4*>(gradp) + i);
      }
    } else {
      reinterpret_cast<f4u*>(p)[i] = pp;
      reinterpret_cast<f4u*>(m)[i] = mm;
      reinterpret_cast<f4u*>(v)[i] = vv;
      if (ZERO_GRAD) {
        f4u zz;
        zz.v[0] = zz.v[1] = zz.v[2] = zz.v[3] = 0.0f;
        reinterpret_cast<f4u*>(gradp)[i] = zz;
      }
    }
  }
  // tail
  const long long i = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float pp = p[i], mm = m[i], vv = v[i];
    adam_update(pp, mm, vv, gradp[i], step, b1, b2, eps, inv_bc2_sqrt);
    p[i] = pp;
    m[i] = mm;
    v[i] = vv;
    if (ZERO_GRAD) gradp[i] = 0.0f;
  }
}

// =============================================================================================
// mean-L1 loss of a rendered batch + its gradient (modules/trainers.py:311-317, 329-336), one launch instead of
// the ~10 elementwise/reduction kernels autograd runs for l1_loss + mse_loss + their backward
// =============================================================================================
struct L1Sets {  // up to two (render, gradient, sums) sets against the same targets in one launch (blockIdx.y)
  const float* colour[2];
  float* grad[2];
  float* sums[2];
};

__device__ __forceinline__ void l1_loss_grad_body(const L1Sets& sets, const float* __restrict__ target, long long n3, float gscale, int set,
                                                  int block, int nblocks) {
  const float* __restrict__ colour = sets.colour[set];
  float* __restrict__ grad = sets.grad[set];
  float* __restrict__ sums = sets.sums[set];
  float abs_sum = 0.0f, sq_sum = 0.0f;
  for (long long i = (long long)block * blockDim.x + threadIdx.x; i < n3; i += (long long)nblocks * blockDim.x) {
    const float d = colour[i] - target[i];
    abs_sum += fabsf(d);
    sq_sum += d * d;
    grad[i] = (d > 0.0f) ? gscale : ((d < 0.0f) ? -gscale : 0.0f);  // sign(d) * scale / numel
  }
  // wave reduce -> block reduce through LDS -> ONE pair of atomics per block (same-address atomics serialise)
  __shared__ float s_part[2][16];  // (up to 1024 threads)
  abs_sum = wave_sum(abs_sum);
  sq_sum = wave_sum(sq_sum);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & (kWave - 1)) == 0) {
    s_part[0][wave] = abs_sum;
    s_part[1][wave] = sq_sum;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.0f, b = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
      a += s_part[0][w];
      b += s_part[1][w];
    }
    unsafeAtomicAdd(sums + 0, a);
    unsafeAtomicAdd(sums + 1, b);
  }
}

__global__ void l1_loss_grad_kernel(L1Sets sets, const float* __restrict__ target, long long n3, float gscale) {
  l1_loss_grad_body(sets, target, n3, gscale, blockIdx.y, blockIdx.x, gridDim.x);
}

// rf_l1_loss_grad_pair: both losses in one launch, finished by the LAST workgroup: the block partial sums go to a workspace (4 sums +
// a ticket counter) with atomics; the workgroup that draws the last ticket reads the totals back (atomics again: the same coherence
// point), writes the five means and leaves the workspace zeroed for the next call.
__global__ void l1_loss_pair_kernel(L1Sets sets, const float* __restrict__ target, long long n3, float gscale, float inv_n3, float* __restrict__ out) {
  l1_loss_grad_body(sets, target, n3, gscale, blockIdx.y, blockIdx.x, gridDim.x);  // (its thread 0 added this block's sums)
  if (threadIdx.x == 0) {
    float* ws = sets.sums[0];  // [0..3] sums (sets.sums[1] = ws + 2), [4] ticket counter
    unsigned int* ticket = reinterpret_cast<unsigned int*>(ws + 4);
    __threadfence();
    const unsigned int total = gridDim.x * gridDim.y;
    if (atomicAdd(ticket, 1u) == total - 1u) {
      __threadfence();
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = atomicExch(ws + i, 0.0f);  // read and clear
      atomicExch(ticket, 0u);
      const float l0 = v[0] * inv_n3, l1 = v[2] * inv_n3;
      out[0] = l0 + l1;
      out[1] = l0;
      out[2] = v[1] * inv_n3;
      out[3] = l1;
      out[4] = v[3] * inv_n3;
    }
  }
}

// rf_train_step: the losses of both renders AND the offsets of both record lists -- everything between the forward passes and the
// adjoints -- in one launch (1024-thread workgroups; blockIdx.y 0, 1: loss of render 0, 1; 2, 3: offsets of list 0, 1)
__global__ __launch_bounds__(1024) void loss_and_offsets_kernel(L1Sets sets, const float* __restrict__ target, long long n3, float gscale,
                                                                int loss_blocks, BinLists lists, int nkeys, int offset_blocks, int passes) {
  if (!((passes >> (blockIdx.y & 1)) & 1)) return;  // (bit i: render i takes part -- a data-parallel caller runs the two renders' chains apart)
  if (blockIdx.y < 2) {
    if ((int)blockIdx.x < loss_blocks) l1_loss_grad_body(sets, target, n3, gscale, blockIdx.y, blockIdx.x, loss_blocks);
  } else if ((int)blockIdx.x < offset_blocks) {
    bin_offsets_body(lists, nkeys, blockIdx.y - 2, blockIdx.x);
  }
}

// ---------------------------------------------------------------------------------------------
// host side of the C ABI
// ---------------------------------------------------------------------------------------------
int check_grid(const RFGrid* g) {
  if (!g || !g->densities_dev) return RF_ERR_NULL_POINTER;
  if (!g->features_dev && !(g->layout != RF_LAYOUT_REFERENCE && g->num_features == 3)) return RF_ERR_NULL_POINTER;
  for (int a = 0; a < 3; ++a)
    if (g->dims[a] < 1 || g->dims[a] > 2046) return RF_ERR_BAD_SHAPE;
  const int F = g->num_features;
  if (!(F == 3 || F == 12 || F == 27 || F == 48)) return RF_ERR_UNSUPPORTED;  // SH degree 0..3
  if (g->density_mode < RF_DENSITY_RELU || g->density_mode > RF_DENSITY_IDENTITY) return RF_ERR_UNSUPPORTED;
  if (g->layout == RF_LAYOUT_REFERENCE) {
    if (g->density_stride < 1 || g->feature_stride < F) return RF_ERR_BAD_SHAPE;
  } else if (g->layout == RF_LAYOUT_SPLIT || g->layout == RF_LAYOUT_BRICKED) {
    if (g->density_stride < 4 || (F > 3 && g->feature_stride < F - 3)) return RF_ERR_BAD_SHAPE;
  } else {
    return RF_ERR_UNSUPPORTED;
  }
  // node indices are 32-bit in the kernels (node_lin: 24-bit multiplies into an unsigned int; element offsets are 64-bit):
  // the node count, padded to whole bricks for the bricked order, must stay below 2^32
  {
    unsigned long long nodes = 1;
    for (int a = 0; a < 3; ++a) nodes *= (unsigned long long)(g->layout == RF_LAYOUT_BRICKED ? (g->dims[a] + 7) / 8 * 8 : g->dims[a]);
    if (nodes >= (1ull << 32)) return RF_ERR_BAD_SHAPE;
  }
  return RF_OK;
}

GridArgs to_args(const RFGrid* g) {
  GridArgs a;
  a.dens = g->densities_dev;
  a.feat = g->features_dev;
  a.occ = g->occupancy_dev;
  a.X = g->dims[0];
  a.Y = g->dims[1];
  a.Z = g->dims[2];
  a.F = g->num_features;
  a.dstride = g->density_stride;
  a.fstride = g->feature_stride;
  for (int i = 0; i < 3; ++i) {
    a.amin[i] = g->aabb_min[i];
    a.amax[i] = g->aabb_max[i];
    a.nscale[i] = g->norm_scale[i];
    a.nbias[i] = g->norm_bias[i];
  }
  a.rho = g->density_scale;
  a.mode = g->density_mode;
  a.layout = g->layout == RF_LAYOUT_REFERENCE ? RF_LAYOUT_REFERENCE : RF_LAYOUT_SPLIT;  // channel arrangement
  a.bricked = g->layout == RF_LAYOUT_BRICKED;
  a.nby = (a.Y + 7) / 8;
  a.nbz = (a.Z + 7) / 8;
  if (a.bricked) {
    a.step[0] = 64u;
    a.step[1] = 8u;
    a.step[2] = 1u;
    a.jump[0] = (unsigned)a.nby * (unsigned)a.nbz * 512u - 7u * 64u;
    a.jump[1] = (unsigned)a.nbz * 512u - 7u * 8u;
    a.jump[2] = 512u - 7u;
  } else {
    a.step[0] = a.jump[0] = (unsigned)a.Y * (unsigned)a.Z;
    a.step[1] = a.jump[1] = (unsigned)a.Z;
    a.step[2] = a.jump[2] = 1u;
  }
  {
    unsigned long long nodes = 1;
    for (int i = 0; i < 3; ++i) nodes *= (unsigned long long)(a.bricked ? (g->dims[i] + 7) / 8 * 8 : g->dims[i]);
    const unsigned long long db = (unsigned long long)a.dstride * 4ull, fb = (unsigned long long)a.fstride * 4ull;
    const uintptr_t d0 = reinterpret_cast<uintptr_t>(a.dens), f0 = a.feat ? reinterpret_cast<uintptr_t>(a.feat) : d0;
    const uintptr_t lo = d0 < f0 ? d0 : f0;
    const unsigned long long dend = (d0 - lo) + nodes * db, fend = a.feat ? (f0 - lo) + nodes * fb : 0ull;
    const unsigned long long span = (dend > fend ? dend : fend) + 16ull;
    a.base = reinterpret_cast<const char*>(lo);
    a.dens_off = (unsigned int)(d0 - lo);
    a.feat_off = (unsigned int)(f0 - lo);
    a.near32 = nodes <= (1ull << 24) && db < (1ull << 24) && fb < (1ull << 24) && span < (1ull << 32);
    // ($RF_FAR_ADDRESSING=1: the general 64-bit path on grids that would not need it -- tests compare the two)
    if (const char* e = getenv("RF_FAR_ADDRESSING")) a.near32 = a.near32 && atoi(e) == 0;
  }
  return a;
}

int check_rays(const RFRayBatch* r) {
  if (!r) return RF_ERR_NULL_POINTER;
  if (r->num_rays < 0 || r->num_samples < 1) return RF_ERR_BAD_SHAPE;
  if (r->num_rays > 0 && !r->t_vals_dev) return RF_ERR_NULL_POINTER;
  if (r->num_rays > 0 && !r->camera && (!r->origins_dev || !r->directions_dev)) return RF_ERR_NULL_POINTER;
  if (r->camera) {
    if (r->camera->height < 1 || r->camera->width < 1 || r->first_ray < 0) return RF_ERR_BAD_SHAPE;
    if (r->first_ray + r->num_rays > (int64_t)r->camera->height * r->camera->width) return RF_ERR_BAD_SHAPE;
  }
  return RF_OK;
}

RayArgs to_args(const RFRayBatch* r, uint32_t flags = 0);
RayArgs to_args(const RFRayBatch* r, uint32_t flags) {
  RayArgs a;
  a.origins = r->origins_dev;
  a.directions = r->directions_dev;
  a.n = r->num_rays;
  a.S = r->num_samples;
  a.near = r->near;
  a.far = r->far;
  a.tvals = r->t_vals_dev;
  a.trand = r->t_rand_dev;
  a.jkey = r->jitter_key;
  a.ray0 = r->first_ray;
  a.jitter = (flags & RF_FLAG_JITTER_KEYED) && !r->t_rand_dev;
  a.cam = r->camera != nullptr;
  a.H = a.W = 1;
  a.focal = 1.0f;
  for (int i = 0; i < 12; ++i) a.pose[i] = 0.0f;
  if (r->camera) {
    a.H = r->camera->height;
    a.W = r->camera->width;
    a.focal = r->camera->focal;
    for (int i = 0; i < 12; ++i) a.pose[i] = r->camera->pose[i];
  }
  return a;
}

OutArgs to_args(const RFRenderOut* o) {
  OutArgs a;
  a.colour = o->colour_dev;
  a.depth = o->depth_dev;
  a.acc = o->acc_dev;
  a.disparity = o->disparity_dev;
  a.cache = o->sample_cache_dev;
  a.tcache = o->trans_cache_dev;
  a.cmask = reinterpret_cast<unsigned long long*>(o->chunk_mask_dev);
  a.stop = o->stop_cache_dev;
  a.hist = nullptr;
  a.brick_shift = 0;
  a.nby = a.nbz = 0;
  return a;
}

int launch_status() { return hipGetLastError() == hipSuccess ? RF_OK : RF_ERR_LAUNCH; }

template <int K, bool DIFFUSE>
void launch_forward(bool save, unsigned blocks, hipStream_t st, const GridArgs& g, const RayArgs& r, const OutArgs& o,
                    uint32_t flags) {
  if (save)
    hipLaunchKernelGGL((render_forward_kernel<K, DIFFUSE, true>), dim3(blocks), dim3(kBlock), 0, st, g, r, o, flags);
  else
    hipLaunchKernelGGL((render_forward_kernel<K, DIFFUSE, false>), dim3(blocks), dim3(kBlock), 0, st, g, r, o, flags);
}

template <int K, bool DIFFUSE>
void launch_backward(unsigned blocks, hipStream_t st, const GridArgs& g, const RayArgs& r, const OutArgs& o,
                     const GradArgs& gr, uint32_t flags) {
  if (gr.sorted)
    hipLaunchKernelGGL((render_emit_direct_kernel<K, DIFFUSE>), dim3(blocks), dim3(kBlock), 0, st, g, r, o, gr, flags);
  else if (gr.keys)
    hipLaunchKernelGGL((render_backward_kernel<K, DIFFUSE, 1>), dim3(blocks), dim3(kBlock), 0, st, g, r, o, gr, flags);
  else
    hipLaunchKernelGGL((render_backward_kernel<K, DIFFUSE, 0>), dim3(blocks), dim3(kBlock), 0, st, g, r, o, gr, flags);
}

unsigned grid_1d(long long n, int block, long long cap = 256LL * 16) {
  long long b = (n + block - 1) / block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (unsigned)b;
}

}  // namespace


extern "C" {

#ifdef RF_BRICK_PROFILE
// development builds only: read (and optionally clear) the phase table of the brick pass; synchronises the device
int rf_debug_brick_profile(unsigned long long* out_host, int reset) {
  if (hipDeviceSynchronize() != hipSuccess) return RF_ERR_LAUNCH;
  if (out_host && hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_brick_prof), sizeof(g_brick_prof)) != hipSuccess) return RF_ERR_LAUNCH;
  if (reset) {
    unsigned long long zero[8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_brick_prof), zero, sizeof(zero)) != hipSuccess) return RF_ERR_LAUNCH;
  }
  return RF_OK;
}
#endif

int rf_abi_version(void) { return RF_ABI_VERSION; }

int rf_abi_struct_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(RFGrid);
    case 1: return (int)sizeof(RFRayBatch);
    case 2: return (int)sizeof(RFRenderOut);
    case 3: return (int)sizeof(RFRenderGrads);
    case 4: return (int)sizeof(RFBrickList);
    case 5: return (int)sizeof(RFAdamState);
    case 6: return (int)sizeof(RFCamera);
    case 7: return (int)sizeof(RFRaySelection);
    case 8: return (int)sizeof(RFPassScratch);
    case 9: return (int)sizeof(RFTrainStep);
    default: return -1;
  }
}

const char* rf_error_string(int code) {
  switch (code) {
    case RF_OK:
      return "ok";
    case RF_ERR_NULL_POINTER:
      return "null pointer";
    case RF_ERR_BAD_SHAPE:
      return "bad shape, size or stride";
    case RF_ERR_UNSUPPORTED:
      return "unsupported configuration (SH degree must be 0..3, density mode a RF_DENSITY_* value, brick size 4 or 8 -- 8 for SH degree 3 -- with at most 4096 bricks for 16-bit keys; the optimizer in the brick flush needs split storage and SH degree 0 or 2)";
    case RF_ERR_LAUNCH:
      return "HIP kernel launch failed";
    default:
      return "unknown error code";
  }
}

int rf_cast_rays(int32_t height, int32_t width, float focal, const float* rotation_host, const float* translation_host,
                 float* origins_dev, float* directions_dev, void* stream) {
  if (!rotation_host || !translation_host || !origins_dev || !directions_dev) return RF_ERR_NULL_POINTER;
  if (height < 1 || width < 1) return RF_ERR_BAD_SHAPE;
  Pose pose;
  for (int i = 0; i < 9; ++i) pose.r[i] = rotation_host[i];
  for (int i = 0; i < 3; ++i) pose.t[i] = translation_host[i];
  const long long n = (long long)height * width;
  hipLaunchKernelGGL(cast_rays_kernel, dim3(grid_1d(n, 256)), dim3(256), 0, (hipStream_t)stream, height, width, focal,
                     pose, origins_dev, directions_dev);
  return launch_status();
}

int rf_cast_selected_rays(int32_t height, int32_t width, float focal, const float* poses_dev, int32_t num_poses,
                          const int64_t* pixel_index_dev, int64_t num_rays, float* origins_dev, float* directions_dev,
                          void* stream) {
  if (num_rays == 0) return RF_OK;
  if (!poses_dev || !pixel_index_dev || !origins_dev || !directions_dev) return RF_ERR_NULL_POINTER;
  if (height < 1 || width < 1 || num_poses < 1 || num_rays < 0) return RF_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(cast_selected_rays_kernel, dim3(grid_1d(num_rays, 256)), dim3(256), 0, (hipStream_t)stream, height,
                     width, focal, poses_dev, num_poses, pixel_index_dev, (long long)num_rays, origins_dev,
                     directions_dev);
  return launch_status();
}

static int select_impl(int32_t height, int32_t width, float focal, const float* poses_dev, const int64_t* image_ids_dev,
                       int32_t num_batch_images, const float* pixel_table_dev, uint64_t key, int64_t first_index, int64_t num_rays,
                       float* origins_dev, float* directions_dev, float* pixels_dev, int64_t* pixel_index_dev, float* zero4_dev,
                       void* stream) {
  if (num_rays == 0) return RF_OK;
  if (!poses_dev || !pixel_table_dev || !origins_dev || !directions_dev || !pixels_dev) return RF_ERR_NULL_POINTER;
  if (height < 1 || width < 1 || num_batch_images < 1 || num_rays < 0 || first_index < 0) return RF_ERR_BAD_SHAPE;
  const unsigned long long P = (unsigned long long)num_batch_images * height * width;
  if ((unsigned long long)(first_index + num_rays) > P) return RF_ERR_BAD_SHAPE;  // cannot draw more distinct pixels than exist
  int bits = 2;
  while ((1ull << bits) < P) ++bits;
  if (bits > 62) return RF_ERR_BAD_SHAPE;
  hipLaunchKernelGGL(select_rays_and_pixels_kernel, dim3(grid_1d(num_rays, 256)), dim3(256), 0, (hipStream_t)stream,
                     height, width, focal, poses_dev, image_ids_dev, num_batch_images, pixel_table_dev,
                     (unsigned long long)key, bits, (long long)first_index, (long long)num_rays, origins_dev, directions_dev,
                     pixels_dev, pixel_index_dev, zero4_dev);
  return launch_status();
}

int rf_select_rays_and_pixels(int32_t height, int32_t width, float focal, const float* poses_dev,
                              const int64_t* image_ids_dev, int32_t num_batch_images, const float* pixel_table_dev,
                              uint64_t key, int64_t first_index, int64_t num_rays, float* origins_dev,
                              float* directions_dev, float* pixels_dev, int64_t* pixel_index_dev, void* stream) {
  return select_impl(height, width, focal, poses_dev, image_ids_dev, num_batch_images, pixel_table_dev, key, first_index, num_rays,
                     origins_dev, directions_dev, pixels_dev, pixel_index_dev, nullptr, stream);
}

int rf_ray_aabb_bounds(const float* origins_dev, const float* directions_dev, int64_t num_rays, float near, float far,
                       const float* aabb_min_host, const float* aabb_max_host, float* bounds_dev, float* hit_dev,
                       void* stream) {
  if (num_rays == 0) return RF_OK;
  if (!origins_dev || !directions_dev || !aabb_min_host || !aabb_max_host || !bounds_dev) return RF_ERR_NULL_POINTER;
  if (num_rays < 0) return RF_ERR_BAD_SHAPE;
  Box box;
  for (int i = 0; i < 3; ++i) {
    box.lo[i] = aabb_min_host[i];
    box.hi[i] = aabb_max_host[i];
  }
  hipLaunchKernelGGL(ray_aabb_bounds_kernel, dim3(grid_1d(num_rays, 256)), dim3(256), 0, (hipStream_t)stream,
                     origins_dev, directions_dev, (long long)num_rays, near, far, box, bounds_dev, hit_dev);
  return launch_status();
}

// short_keys: the (brick, flags) key must fit a positive 16-bit sort key (per-slot key arrays of rf_render_backward_emit);
// the fused binning only needs the counters to be addressable (2^21 keys = grids up to 512^3 at 8^3 bricks)
// *shift: log2 of the brick edge; for the 4 x 8 x 8 bricks of RF_BRICK_4X8X8 bit 4 is set as well (brick_key: the x edge is half as long)
static int brick_geometry(const RFGrid* grid, int brick_size, int* shift, int nb[3], bool short_keys) {
  if (brick_size != 4 && brick_size != 8 && brick_size != RF_BRICK_4X8X8) return RF_ERR_UNSUPPORTED;
  const bool slab4 = brick_size == RF_BRICK_4X8X8;
  *shift = (brick_size == 4) ? 2 : (slab4 ? (3 | (1 << 4)) : 3);
  long long total = 1;
  for (int a = 0; a < 3; ++a) {
    const int edge = slab4 ? (a == 0 ? 4 : 8) : brick_size;
    nb[a] = (grid->dims[a] + edge - 1) / edge;
    total *= nb[a];
  }
  return (total * 8 - 1 <= (short_keys ? 0x7fffLL : (1LL << 21) - 1)) ? RF_OK : RF_ERR_UNSUPPORTED;
}



// Which kernel renders a frame of a posed camera (RFRayBatch.camera, no sample cache): ray packets -- one wave per 8 x 8 pixel tile,
// render_frame_tile_kernel -- where a tile's rays stay within ~2 voxels of each other at the volume's centre (8 pixels x distance /
// focal length against the smallest voxel edge; 3 voxels with the occupancy mask, whose live lanes are few): beyond that the
// 4 x 4 x 4-node window has to be moved several times per step and the per-ray kernel wins.  Measured, 800 x 800: 128^3 / 256 samples
// (1.3 voxels) 1.78 against 2.80 ms; 256^3 / 512 samples (2.6 voxels) 1.70 against 2.05 ms with the mask, 3.12 against 2.90 ms without.
// The packet kernel exists for split / bricked storage (a grid in the reference's tensors is rendered from its split shadow), every SH degree.  $RF_FRAME_TILES = 1 / 0 forces /
// forbids it where it exists (A/B runs, tests).
constexpr int kTileMaxSamples = 1024;  // (the reference's default render_num_samples_per_ray: the largest count the sweep holds the packet kernel to the bar at)
static bool frame_uses_packets(const RFGrid* grid, const GridArgs& g, const RFCamera* cam, uint32_t flags) {
  const int K = grid->num_features / 3;
  if (!(g.layout == RF_LAYOUT_SPLIT && (K == 1 || K == 4 || K == 9 || K == 16) && g.Z >= 4 && g.Y >= 4 && g.X >= 4 && (g.dstride & 3) == 0 && (K != 9 || (g.fstride & 3) == 0))) return false;
  // (the window is fetched as 16-byte quads: the base tensor -- and the degree-2 rest tensor -- must start on a 16-byte boundary)
  if ((((uintptr_t)grid->densities_dev | (uintptr_t)(K == 9 ? grid->features_dev : nullptr)) & 15u) != 0) return false;
  if (const char* e = getenv("RF_FRAME_TILES")) return atoi(e) != 0;
  float dist2 = 0.0f, vmin = 1e30f;
  for (int a = 0; a < 3; ++a) {
    const float c = 0.5f * (g.amin[a] + g.amax[a]) - cam->pose[4 * a + 3];
    dist2 += c * c;
    const int dim = a == 0 ? g.X : (a == 1 ? g.Y : g.Z);
    vmin = fminf(vmin, (g.amax[a] - g.amin[a]) / (float)dim);
  }
  return 8.0f * sqrtf(dist2) / cam->focal <= (((flags & RF_FLAG_OCCUPANCY_SKIP) && grid->occupancy_dev) ? 3.0f : 2.0f) * vmin;
}

int rf_frame_render_kernel(const RFGrid* grid, const RFCamera* camera, uint32_t flags) {
  int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  if (!camera) return RF_ERR_NULL_POINTER;
  if (camera->height < 1 || camera->width < 1 || !(camera->focal > 0.0f)) return RF_ERR_BAD_SHAPE;
  return frame_uses_packets(grid, to_args(grid), camera, flags) ? 1 : 0;
}

int rf_render_forward(const RFGrid* grid, const RFRayBatch* rays, uint32_t flags, const RFRenderOut* out,
                      void* stream) {
  int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  rc = check_rays(rays);
  if (rc != RF_OK) return rc;
  if (!out) return RF_ERR_NULL_POINTER;
  if (rays->num_rays == 0) return RF_OK;
  if (!out->colour_dev || !out->depth_dev || !out->acc_dev || !out->disparity_dev) return RF_ERR_NULL_POINTER;
  const bool save = out->sample_cache_dev != nullptr;
  if (save && (!out->trans_cache_dev || !out->stop_cache_dev || !out->chunk_mask_dev)) return RF_ERR_NULL_POINTER;
  if ((flags & RF_FLAG_OCCUPANCY_SKIP) && !grid->occupancy_dev) return RF_ERR_NULL_POINTER;

  const GridArgs g = to_args(grid);
  const RayArgs r = to_args(rays, flags);
  OutArgs o = to_args(out);
  if (out->key_hist_dev) {  // count the records of the binned backward per (brick, flags) key
    if (!save) return RF_ERR_NULL_POINTER;
    int shift, nb[3];
    rc = brick_geometry(grid, out->brick_size, &shift, nb, false);
    if (rc != RF_OK) return rc;
    o.hist = out->key_hist_dev;
    o.brick_shift = shift;
    o.nby = nb[1];
    o.nbz = nb[2];
  }
  const unsigned blocks = (unsigned)((rays->num_rays + kWavesPerBlock - 1) / kWavesPerBlock);
  hipStream_t st = (hipStream_t)stream;
  const bool diffuse = flags & RF_FLAG_RENDER_DIFFUSE;
  const int K = grid->num_features / 3;
  // frames of a posed camera (rays generated in-kernel, inference only): ray packets where frame_uses_packets() says so
  if (rays->camera && !save) {
    // (a packet lane adds its ray's weighted samples one after the other; beyond kTileMaxSamples samples per ray that sequential float32
    // sum drifts past the parity bar -- 3..8e-5 on depth at 4096+ samples, where torch.sum's pairwise order keeps the reference within
    // 2e-6 -- so such frames go to the per-ray kernel, whose per-lane sums are 64 times shorter, whatever $RF_FRAME_TILES says)
    const bool tiles = !rays->t_rand_dev && rays->num_samples <= kTileMaxSamples && frame_uses_packets(grid, g, rays->camera, flags);
    if (tiles) {
      const int W = rays->camera->width;
      const int row0 = (int)(rays->first_ray / W), row1 = (int)((rays->first_ray + rays->num_rays - 1) / W);
      const int tile_rows = (row1 - row0) / 8 + 1, tiles_x = (W + 7) / 8;
      // (scheduling of the tiles: $RF_TILE_WPB = waves per workgroup, 1 or 4; $RF_TILE_XCD_ROWS = 0 / 1: A/B switches, read once)
      static const int wpb = [] {
        const char* e = getenv("RF_TILE_WPB");
        return (e && atoi(e) == 4) ? 4 : 1;
      }();
      static const bool xcd_rows = [] {
        const char* e = getenv("RF_TILE_XCD_ROWS");
        return e ? atoi(e) != 0 : true;
      }();
      const int KT = diffuse ? 1 : K;  // coefficients per colour that are read
      const bool sched_default = KT == 4 || KT == 16;  // (the degree-1 / 3 instantiations exist in the default scheduling only)
      const int wpb_ = sched_default ? 1 : wpb;
      const bool xr_ = sched_default ? true : xcd_rows;
      const long long wgs = xr_ ? (long long)((tile_rows + 7) / 8) * 8 * ((tiles_x + wpb_ - 1) / wpb_) : ((long long)tile_rows * tiles_x + wpb_ - 1) / wpb_;
#define RF_TILE_LAUNCH(K_, WPB_, XR_)                                                                                                            \
  hipLaunchKernelGGL((render_frame_tile_kernel<K_, WPB_, XR_>), dim3((unsigned)wgs), dim3(kWave * WPB_), 0, st, g, r, o, flags, row0, tile_rows, tiles_x)
      if (KT == 4) {
        RF_TILE_LAUNCH(4, 1, true);
      } else if (KT == 16) {
        RF_TILE_LAUNCH(16, 1, true);
      } else if (KT == 9) {
        if (wpb == 4) { if (xcd_rows) RF_TILE_LAUNCH(9, 4, true); else RF_TILE_LAUNCH(9, 4, false); }
        else { if (xcd_rows) RF_TILE_LAUNCH(9, 1, true); else RF_TILE_LAUNCH(9, 1, false); }
      } else {
        if (wpb == 4) { if (xcd_rows) RF_TILE_LAUNCH(1, 4, true); else RF_TILE_LAUNCH(1, 4, false); }
        else { if (xcd_rows) RF_TILE_LAUNCH(1, 1, true); else RF_TILE_LAUNCH(1, 1, false); }
      }
#undef RF_TILE_LAUNCH
      return launch_status();
    }
  }
  if (diffuse || K == 1)
    launch_forward<1, true>(save, blocks, st, g, r, o, flags);
  else if (K == 4)
    launch_forward<4, false>(save, blocks, st, g, r, o, flags);
  else if (K == 9)
    launch_forward<9, false>(save, blocks, st, g, r, o, flags);
  else
    launch_forward<16, false>(save, blocks, st, g, r, o, flags);
  return launch_status();
}

// both saving forward renders of a training iteration (0 = specular, 1 = render_diffuse) over the same rays in ONE launch
// (render_forward_pair_kernel); RF_ERR_UNSUPPORTED when the two calls do not pair up -- the caller then launches them one by one
static int forward_pair_impl(const RFGrid* grid, const RFRayBatch* const rays[2], const uint32_t flags[2], const RFRenderOut* const outs[2], void* stream) {
  int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  ForwardPair p;
  for (int i = 0; i < 2; ++i) {
    rc = check_rays(rays[i]);
    if (rc != RF_OK) return rc;
    const RFRenderOut* out = outs[i];
    if (!out || !out->colour_dev || !out->depth_dev || !out->acc_dev || !out->disparity_dev) return RF_ERR_NULL_POINTER;
    if (!out->sample_cache_dev || !out->trans_cache_dev || !out->stop_cache_dev || !out->chunk_mask_dev) return RF_ERR_UNSUPPORTED;
    if ((flags[i] & RF_FLAG_OCCUPANCY_SKIP) && !grid->occupancy_dev) return RF_ERR_NULL_POINTER;
    p.r[i] = to_args(rays[i], flags[i]);
    p.out[i] = to_args(out);
    p.flags[i] = flags[i];
    if (out->key_hist_dev) {
      int shift, nb[3];
      rc = brick_geometry(grid, out->brick_size, &shift, nb, false);
      if (rc != RF_OK) return rc;
      p.out[i].hist = out->key_hist_dev;
      p.out[i].brick_shift = shift;
      p.out[i].nby = nb[1];
      p.out[i].nbz = nb[2];
    }
  }
  if (rays[0]->num_rays != rays[1]->num_rays || (flags[0] & RF_FLAG_RENDER_DIFFUSE) || !(flags[1] & RF_FLAG_RENDER_DIFFUSE)) return RF_ERR_UNSUPPORTED;
  if (rays[0]->num_rays == 0) return RF_OK;
  const GridArgs g = to_args(grid);
  const unsigned blocks = (unsigned)((rays[0]->num_rays + 1) / 2);
  hipStream_t st = (hipStream_t)stream;
  switch (grid->num_features / 3) {
    case 1: hipLaunchKernelGGL((render_forward_pair_kernel<1>), dim3(blocks), dim3(kBlock), 0, st, g, p); break;
    case 4: hipLaunchKernelGGL((render_forward_pair_kernel<4>), dim3(blocks), dim3(kBlock), 0, st, g, p); break;
    case 9: hipLaunchKernelGGL((render_forward_pair_kernel<9>), dim3(blocks), dim3(kBlock), 0, st, g, p); break;
    default: hipLaunchKernelGGL((render_forward_pair_kernel<16>), dim3(blocks), dim3(kBlock), 0, st, g, p); break;
  }
  return launch_status();
}

static int backward_impl(const RFGrid* grid, const RFRayBatch* rays, uint32_t flags, const RFRenderOut* fwd,
                         const RFRenderGrads* grads, GradArgs gr, void* stream) {
  int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  rc = check_rays(rays);
  if (rc != RF_OK) return rc;
  if (!fwd || !grads) return RF_ERR_NULL_POINTER;
  if (rays->num_rays == 0) return RF_OK;
  if (!fwd->sample_cache_dev || !fwd->trans_cache_dev || !fwd->stop_cache_dev || !fwd->chunk_mask_dev) return RF_ERR_NULL_POINTER;
  const GridArgs g = to_args(grid);
  const RayArgs r = to_args(rays, flags);
  const OutArgs o = to_args(fwd);
  gr.gcolour = grads->grad_colour_dev;
  gr.gdepth = grads->grad_depth_dev;
  gr.gacc = grads->grad_acc_dev;
  const unsigned blocks = (unsigned)((rays->num_rays + kWavesPerBlock - 1) / kWavesPerBlock);
  hipStream_t st = (hipStream_t)stream;
  const bool diffuse = flags & RF_FLAG_RENDER_DIFFUSE;
  const int K = grid->num_features / 3;
  if (diffuse || K == 1)
    launch_backward<1, true>(blocks, st, g, r, o, gr, flags);
  else if (K == 4)
    launch_backward<4, false>(blocks, st, g, r, o, gr, flags);
  else if (K == 9)
    launch_backward<9, false>(blocks, st, g, r, o, gr, flags);
  else
    launch_backward<16, false>(blocks, st, g, r, o, gr, flags);
  return launch_status();
}

int rf_render_backward(const RFGrid* grid, const RFRayBatch* rays, uint32_t flags, const RFRenderOut* fwd,
                       const RFRenderGrads* grads, float* grad_densities_dev, float* grad_features_dev, void* stream) {
  if (!grad_densities_dev) return RF_ERR_NULL_POINTER;
  if (grid && !grad_features_dev && !(grid->layout != RF_LAYOUT_REFERENCE && grid->num_features == 3)) return RF_ERR_NULL_POINTER;
  GradArgs gr = {};
  gr.gdens = grad_densities_dev;
  gr.gfeat = grad_features_dev;
  return backward_impl(grid, rays, flags, fwd, grads, gr, stream);
}


int rf_render_backward_emit(const RFGrid* grid, const RFRayBatch* rays, uint32_t flags, const RFRenderOut* fwd,
                            const RFRenderGrads* grads, int32_t brick_size, int16_t* keys_dev, float* records_dev,
                            int32_t* hist_dev, void* stream) {
  if (!grid) return RF_ERR_NULL_POINTER;
  if (!keys_dev || !records_dev) return RF_ERR_NULL_POINTER;
  int shift, nb[3];
  const int rc = brick_geometry(grid, brick_size, &shift, nb, true);
  if (rc != RF_OK) return rc;
  GradArgs gr = {};
  gr.keys = keys_dev;
  gr.records = records_dev;
  gr.brick_shift = shift;
  gr.nby = nb[1];
  gr.nbz = nb[2];
  gr.hist = hist_dev;
  return backward_impl(grid, rays, flags, fwd, grads, gr, stream);
}

// the direct-emit adjoints of both renders of a training iteration (0 = specular, 1 = render_diffuse) in ONE launch
static int emit_pair_impl(const RFGrid* grid, const RFRayBatch* const rays[2], const uint32_t flags[2], const RFPassScratch* const pass[2], void* stream) {
  int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  EmitPair p;
  for (int i = 0; i < 2; ++i) {
    rc = check_rays(rays[i]);
    if (rc != RF_OK) return rc;
    const RFPassScratch& ps = *pass[i];
    const RFRenderOut* fwd = &ps.out;
    if (!ps.cursor_dev || !ps.records_sorted_dev || !ps.grad_colour_dev) return RF_ERR_NULL_POINTER;
    if (!fwd->sample_cache_dev || !fwd->trans_cache_dev || !fwd->stop_cache_dev || !fwd->chunk_mask_dev) return RF_ERR_NULL_POINTER;
    int shift, nb[3];
    rc = brick_geometry(grid, fwd->brick_size, &shift, nb, false);
    if (rc != RF_OK) return rc;
    p.r[i] = to_args(rays[i], flags[i]);
    p.fwd[i] = to_args(fwd);
    p.flags[i] = flags[i];
    GradArgs gr = {};
    gr.cursor = ps.cursor_dev;
    gr.sorted = reinterpret_cast<float4*>(ps.records_sorted_dev);
    gr.hist_clear = fwd->key_hist_dev;
    gr.nkeys = nb[0] * nb[1] * nb[2] * 8;
    gr.brick_shift = shift;
    gr.nby = nb[1];
    gr.nbz = nb[2];
    gr.gcolour = ps.grad_colour_dev;
    p.gr[i] = gr;
  }
  if (rays[0]->num_rays != rays[1]->num_rays || (flags[0] & RF_FLAG_RENDER_DIFFUSE) || !(flags[1] & RF_FLAG_RENDER_DIFFUSE)) return RF_ERR_UNSUPPORTED;
  if (rays[0]->num_rays == 0) return RF_OK;
  const GridArgs g = to_args(grid);
  const unsigned blocks = (unsigned)((rays[0]->num_rays + 1) / 2);
  hipStream_t st = (hipStream_t)stream;
  switch (grid->num_features / 3) {
    case 1: hipLaunchKernelGGL((render_emit_direct_pair_kernel<1>), dim3(blocks), dim3(kBlock), 0, st, g, p); break;
    case 4: hipLaunchKernelGGL((render_emit_direct_pair_kernel<4>), dim3(blocks), dim3(kBlock), 0, st, g, p); break;
    case 9: hipLaunchKernelGGL((render_emit_direct_pair_kernel<9>), dim3(blocks), dim3(kBlock), 0, st, g, p); break;
    default: hipLaunchKernelGGL((render_emit_direct_pair_kernel<16>), dim3(blocks), dim3(kBlock), 0, st, g, p); break;
  }
  return launch_status();
}

int rf_render_forward_pair(const RFGrid* grid, const RFRayBatch* rays, const uint32_t* flags, const RFRenderOut* outs, void* stream) {
  if (!grid || !rays || !flags || !outs) return RF_ERR_NULL_POINTER;
  const RFRayBatch* const rr[2] = {&rays[0], &rays[1]};
  const RFRenderOut* const oo[2] = {&outs[0], &outs[1]};
  if (rays[0].camera || rays[1].camera) return RF_ERR_UNSUPPORTED;  // (ray lists: the renders of a training iteration)
  return forward_pair_impl(grid, rr, flags, oo, stream);
}

int rf_render_backward_emit_direct_pair(const RFGrid* grid, const RFRayBatch* rays, const uint32_t* flags, const RFPassScratch* passes, void* stream) {
  if (!grid || !rays || !flags || !passes) return RF_ERR_NULL_POINTER;
  if (passes[0].out.brick_size != passes[1].out.brick_size) return RF_ERR_BAD_SHAPE;
  const RFRayBatch* const rr[2] = {&rays[0], &rays[1]};
  const RFPassScratch* const pp[2] = {&passes[0], &passes[1]};
  return emit_pair_impl(grid, rr, flags, pp, stream);
}

int rf_render_backward_emit_direct(const RFGrid* grid, const RFRayBatch* rays, uint32_t flags, const RFRenderOut* fwd,
                                   const RFRenderGrads* grads, int32_t brick_size, int32_t* cursor_dev,
                                   float* records_sorted_dev, int32_t* hist_clear_dev, void* stream) {
  if (!grid) return RF_ERR_NULL_POINTER;
  if (!cursor_dev || !records_sorted_dev) return RF_ERR_NULL_POINTER;
  int shift, nb[3];
  const int rc = brick_geometry(grid, brick_size, &shift, nb, false);
  if (rc != RF_OK) return rc;
  GradArgs gr = {};
  gr.cursor = cursor_dev;
  gr.sorted = reinterpret_cast<float4*>(records_sorted_dev);
  gr.hist_clear = hist_clear_dev;
  gr.nkeys = nb[0] * nb[1] * nb[2] * 8;
  gr.brick_shift = shift;
  gr.nby = nb[1];
  gr.nbz = nb[2];
  return backward_impl(grid, rays, flags, fwd, grads, gr, stream);
}

extern "C++" {
template <int Q>
static int launch_expand(const float* records_dev, const int64_t* perm_dev, const int64_t* begin_dev, int64_t capacity, float* out, hipStream_t st) {
  hipLaunchKernelGGL((expand_records_kernel<Q>), dim3(grid_1d(capacity * Q, 256, 256LL * 16)), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(records_dev), reinterpret_cast<const long long*>(perm_dev),
                     reinterpret_cast<const long long*>(begin_dev), (long long)capacity, reinterpret_cast<float4*>(out));
  return launch_status();
}
}  // extern "C++"

int rf_expand_records(const RFGrid* grid, const float* records_dev, const int64_t* perm_dev, const int64_t* begin_dev,
                      int64_t capacity, int32_t render_diffuse, float* records_sorted_dev, void* stream) {
  const int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  if (capacity == 0) return RF_OK;
  if (!records_dev || !perm_dev || !begin_dev || !records_sorted_dev) return RF_ERR_NULL_POINTER;
  if (capacity < 0) return RF_ERR_BAD_SHAPE;
  const int diffuse = render_diffuse || grid->num_features == 3;
  hipStream_t st = (hipStream_t)stream;
  if (diffuse) return launch_expand<2>(records_dev, perm_dev, begin_dev, capacity, records_sorted_dev, st);  // base-channel records
  return launch_expand<3>(records_dev, perm_dev, begin_dev, capacity, records_sorted_dev, st);
}

static int bin_offsets_impl(const int32_t* const hist[2], int64_t* const offsets[2], int32_t* const cursor[2], int nlists, int32_t num_keys,
                            void* stream) {
  if (num_keys < 1 || num_keys > (1 << 21)) return RF_ERR_BAD_SHAPE;
  BinLists l = {};
  for (int i = 0; i < nlists; ++i) {
    if (!hist[i] || !offsets[i] || !cursor[i]) return RF_ERR_NULL_POINTER;
    l.hist[i] = hist[i];
    l.offsets[i] = reinterpret_cast<long long*>(offsets[i]);
    l.cursor[i] = cursor[i];
  }
  hipLaunchKernelGGL(bin_offsets_kernel, dim3((num_keys + 1023) / 1024, nlists), dim3(1024), 0, (hipStream_t)stream, l, num_keys);
  return launch_status();
}

int rf_bin_offsets_pair(const int32_t* const* hist_dev, int32_t num_keys, int64_t* const* offsets_dev, int32_t* const* cursor_dev, void* stream) {
  if (!hist_dev || !offsets_dev || !cursor_dev) return RF_ERR_NULL_POINTER;
  const int32_t* h[2] = {hist_dev[0], hist_dev[1]};
  int64_t* o[2] = {offsets_dev[0], offsets_dev[1]};
  int32_t* c[2] = {cursor_dev[0], cursor_dev[1]};
  return bin_offsets_impl(h, o, c, 2, num_keys, stream);
}

int rf_bin_offsets(const int32_t* hist_dev, int32_t num_keys, int64_t* offsets_dev, int32_t* cursor_dev, void* stream) {
  const int32_t* h[2] = {hist_dev, nullptr};
  int64_t* o[2] = {offsets_dev, nullptr};
  int32_t* c[2] = {cursor_dev, nullptr};
  return bin_offsets_impl(h, o, c, 1, num_keys, stream);
}

extern "C++" {
template <int Q>
static int launch_scatter(const int16_t* keys_dev, const float* records_dev, int64_t capacity, int32_t* cursor_dev, float* out,
                          int32_t* hist_dev, int nkeys, hipStream_t st) {
  hipLaunchKernelGGL((scatter_records_kernel<Q>), dim3((unsigned)((capacity + 1023) / 1024)), dim3(256), 0, st, keys_dev,
                     reinterpret_cast<const float4*>(records_dev), (long long)capacity, cursor_dev, reinterpret_cast<float4*>(out), hist_dev, nkeys);
  return launch_status();
}
}  // extern "C++"

int rf_scatter_records(const RFGrid* grid, const int16_t* keys_dev, const float* records_dev, int64_t capacity,
                       int32_t* cursor_dev, int32_t render_diffuse, float* records_sorted_dev, int32_t* hist_dev, int32_t num_keys,
                       void* stream) {
  const int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  if (capacity == 0) return RF_OK;
  if (!keys_dev || !records_dev || !cursor_dev || !records_sorted_dev) return RF_ERR_NULL_POINTER;
  if (capacity < 0) return RF_ERR_BAD_SHAPE;
  const int diffuse = render_diffuse || grid->num_features == 3;
  hipStream_t st = (hipStream_t)stream;
  if (diffuse) return launch_scatter<2>(keys_dev, records_dev, capacity, cursor_dev, records_sorted_dev, hist_dev, num_keys, st);
  return launch_scatter<3>(keys_dev, records_dev, capacity, cursor_dev, records_sorted_dev, hist_dev, num_keys, st);
}

int32_t rf_expanded_record_floats(int32_t num_features) { return 4 * record_quads(num_features / 3); }

extern "C++" {
template <int K, bool ADAM, bool ONE_ROUND = false, bool SPLIT = false, int BX = 8, bool MIRROR = false>
static int launch_gather(const GridArgs& g, const BrickArgs& a, int nbricks, float* gd, float* gf, hipStream_t st) {
  const int B = 1 << a.shift;
  const size_t lds = (size_t)gather_lds_words(B, 3 * K + 1, BX) * sizeof(float);
  if (lds > 150 * 1024) return RF_ERR_UNSUPPORTED;
  static std::atomic<size_t> configured[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return RF_ERR_LAUNCH;
  if (lds > configured[dev].load(std::memory_order_relaxed)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&brick_gather_kernel<K, ADAM, ONE_ROUND, SPLIT, BX, MIRROR>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return RF_ERR_LAUNCH;
    configured[dev].store(lds, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL((brick_gather_kernel<K, ADAM, ONE_ROUND, SPLIT, BX, MIRROR>), dim3(nbricks * (SPLIT ? a.parts : 1)), dim3(BX == 4 ? kBrickThreads / 2 : kBrickThreads), lds, st, g, a,
                     gd, gf);
  return launch_status();
}
}  // extern "C++"

// what the optimizer in the brick flush needs of the grid and the state (checked before anything is launched)
static int check_fused_adam(const RFGrid* grid, const RFAdamState* adam, int K, bool base_only) {
  // the update needs the complete gradient of every parameter in the owning workgroup: all channels covered by the lists
  // (base-only lists on an SH grid leave the higher-degree channels to someone else), whole float4s, overwrite semantics
  const int C = 3 * K + 1;
  if (grid->layout == RF_LAYOUT_REFERENCE || (C & 3) || (base_only && grid->num_features != 3)) return RF_ERR_UNSUPPORTED;
  if ((grid->density_stride & 3) || (C > 4 && (grid->feature_stride & 3))) return RF_ERR_UNSUPPORTED;
  if (!adam->param_first_dev || !adam->exp_avg_first_dev || !adam->exp_avg_sq_first_dev) return RF_ERR_NULL_POINTER;
  if (C > 4 && (!adam->param_second_dev || !adam->exp_avg_second_dev || !adam->exp_avg_sq_second_dev)) return RF_ERR_NULL_POINTER;
  if (adam->param_first_dev != grid->densities_dev || (C > 4 && adam->param_second_dev != grid->features_dev)) return RF_ERR_BAD_SHAPE;
  if (adam->step < 1) return RF_ERR_BAD_SHAPE;
  {  // the flush keeps element offsets in 31 bits
    unsigned long long nodes = 1;
    for (int ax = 0; ax < 3; ++ax) nodes *= (unsigned long long)((grid->dims[ax] + 7) / 8 * 8);
    const unsigned long long smax = (unsigned long long)(grid->density_stride > grid->feature_stride ? grid->density_stride : grid->feature_stride);
    if (nodes * smax >= (1ull << 31)) return RF_ERR_UNSUPPORTED;
  }
  const uintptr_t align = (uintptr_t)adam->param_first_dev | (uintptr_t)adam->exp_avg_first_dev | (uintptr_t)adam->exp_avg_sq_first_dev |
                          (C > 4 ? ((uintptr_t)adam->param_second_dev | (uintptr_t)adam->exp_avg_second_dev | (uintptr_t)adam->exp_avg_sq_second_dev) : 0);
  if (align & 15u) return RF_ERR_BAD_SHAPE;
  return RF_OK;
}

static int brick_accumulate_impl(const RFGrid* grid, int32_t brick_size, const RFBrickList* lists, int32_t num_lists,
                                 float* grad_densities_dev, float* grad_features_dev, int32_t accumulate,
                                 const RFAdamState* adam, int32_t first_brick, int32_t num_bricks, void* stream, int32_t parts = 1,
                                 void* scratch_dev = nullptr, int64_t scratch_bytes = 0) {
  int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  if (!lists) return RF_ERR_NULL_POINTER;
  if (!adam) {
    if (!grad_densities_dev) return RF_ERR_NULL_POINTER;
    if (!grad_features_dev && !(grid->layout != RF_LAYOUT_REFERENCE && grid->num_features == 3)) return RF_ERR_NULL_POINTER;
  }
  if (num_lists < 1 || num_lists > 2 * kMaxListsPerKind) return RF_ERR_BAD_SHAPE;
  int shift, nb[3];
  rc = brick_geometry(grid, brick_size, &shift, nb, false);
  if (rc != RF_OK) return rc;
  const bool slab4 = (shift >> 4) != 0;  // RF_BRICK_4X8X8
  shift &= 15;
  // lists: the full-width ones first, then the base-channel lists of render_diffuse passes (on a degree-0 grid, or when every list
  // is a render_diffuse list, there is one kind only and the pass runs on the 4 base channels)
  BrickArgs a = {};
  int ndiffuse = 0;
  for (int i = 0; i < num_lists; ++i) {
    if (!lists[i].records_sorted_dev || !lists[i].offsets_dev) return RF_ERR_NULL_POINTER;
    const bool diffuse = lists[i].render_diffuse || grid->num_features == 3;
    if (!diffuse && ndiffuse) return RF_ERR_BAD_SHAPE;  // the specular lists come first
    ndiffuse += diffuse;
  }
  const bool base_only = ndiffuse == num_lists;  // 4 accumulator channels per node, written to the base channels only
  for (int i = 0; i < num_lists; ++i) {
    const bool narrow = !base_only && (lists[i].render_diffuse != 0);
    int& n = narrow ? a.nnarrow : a.nwide;
    if (n >= kMaxListsPerKind) return RF_ERR_BAD_SHAPE;
    BrickList& dst = narrow ? a.narrow[n] : a.wide[n];
    dst.rec = reinterpret_cast<const float4*>(lists[i].records_sorted_dev);
    dst.offsets = reinterpret_cast<const long long*>(lists[i].offsets_dev);
    ++n;
  }
  a.fmul = base_only ? grid->num_features / 3 : 1;
  a.shift = shift;
  a.nbx = nb[0];
  a.nby = nb[1];
  a.nbz = nb[2];
  a.accumulate = accumulate;
  const GridArgs g = to_args(grid);
  const int all_bricks = nb[0] * nb[1] * nb[2];
  if (first_brick < 0 || num_bricks < 0 || first_brick + num_bricks > all_bricks) return RF_ERR_BAD_SHAPE;
  const int nbricks = num_bricks > 0 ? num_bricks : all_bricks - first_brick;
  a.brick_first = first_brick;
  if (nbricks == 0) return RF_OK;
  hipStream_t st = (hipStream_t)stream;
  const int K = base_only ? 1 : grid->num_features / 3;
  if (adam) {
    if (accumulate) return RF_ERR_UNSUPPORTED;
    rc = check_fused_adam(grid, adam, K, base_only);
    if (rc != RF_OK) return rc;
    const double bc1 = 1.0 - pow((double)adam->beta1, (double)adam->step);
    const double bc2 = 1.0 - pow((double)adam->beta2, (double)adam->step);
    a.adam.p1 = adam->param_first_dev;
    a.adam.p2 = adam->param_second_dev;
    a.adam.m1 = adam->exp_avg_first_dev;
    a.adam.m2 = adam->exp_avg_second_dev;
    a.adam.v1 = adam->exp_avg_sq_first_dev;
    a.adam.v2 = adam->exp_avg_sq_second_dev;
    a.adam.step = adam->lr / (float)bc1;  // (float division, like rf_adam_step's kernel: step = lr / bc1)
    a.adam.b1 = adam->beta1;
    a.adam.b2 = adam->beta2;
    a.adam.eps = adam->eps;
    a.adam.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    {
      unsigned long long nodes = 1;
      for (int ax = 0; ax < 3; ++ax) nodes *= (unsigned long long)((grid->dims[ax] + 7) / 8 * 8);
      const unsigned long long smax = (unsigned long long)(grid->density_stride > grid->feature_stride ? grid->density_stride : grid->feature_stride);
      a.adam.byte_offsets_fit_32_bits = nodes * smax < (1ull << 30) && nodes <= (1ull << 24) && smax < (1ull << 24);
    }
#if defined(RF_BRICK_PROFILE) || defined(RF_BRICK_ABLATE)
    {
      const char* e = getenv("RF_BRICK_STAGGER");
      a.stagger = e ? atoi(e) : 0;
    }
#endif
    const bool one_round = a.adam.byte_offsets_fit_32_bits && shift == 3;
    if (slab4) {  // 4 x 8 x 8 bricks: the 256-thread workgroups of the single-GPU optimizer pass
      if (!one_round || parts > 1 || (K != 1 && K != 9)) return RF_ERR_UNSUPPORTED;
      if (grad_densities_dev) {  // rf_brick_accumulate_adam_mirror: the updated parameters also in the reference layout
        if (!grad_features_dev || (grid->dims[0] & 3) || (grid->dims[1] & 7) || (grid->dims[2] & 7) || g.bricked || first_brick != 0 || nbricks != nb[0] * nb[1] * nb[2])
          return RF_ERR_UNSUPPORTED;
        if ((reinterpret_cast<uintptr_t>(grad_densities_dev) | reinterpret_cast<uintptr_t>(grad_features_dev)) & 15u) return RF_ERR_BAD_SHAPE;
        return K == 1 ? launch_gather<1, true, true, false, 4, true>(g, a, nbricks, grad_densities_dev, grad_features_dev, st)
                      : launch_gather<9, true, true, false, 4, true>(g, a, nbricks, grad_densities_dev, grad_features_dev, st);
      }
      return K == 1 ? launch_gather<1, true, true, false, 4>(g, a, nbricks, nullptr, nullptr, st) : launch_gather<9, true, true, false, 4>(g, a, nbricks, nullptr, nullptr, st);
    }
    if (grad_densities_dev) return RF_ERR_UNSUPPORTED;  // (the mirror write-out exists for the 4 x 8 x 8 pass only)
    if (parts > 1) {  // several workgroups per brick (rf_brick_accumulate_adam_split)
      if (!one_round) return RF_ERR_UNSUPPORTED;
      if (parts > kMaxListsPerKind || !scratch_dev) return parts > kMaxListsPerKind ? RF_ERR_BAD_SHAPE : RF_ERR_NULL_POINTER;
      const long long words = brick_acc_words(8, 3 * K + 1);
      const long long state_bytes = ((long long)nbricks * (1 + parts) * 4 + 255) / 256 * 256;
      if (scratch_bytes < state_bytes + (long long)nbricks * parts * words * 4 || ((uintptr_t)scratch_dev & 15u)) return RF_ERR_BAD_SHAPE;
      a.parts = parts;
      a.part_state = reinterpret_cast<int*>(scratch_dev);
      a.partial = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch_dev) + state_bytes);
      return K == 1 ? launch_gather<1, true, true, true>(g, a, nbricks, nullptr, nullptr, st) : launch_gather<9, true, true, true>(g, a, nbricks, nullptr, nullptr, st);
    }
    switch (K) {
      case 1:
        return one_round ? launch_gather<1, true, true>(g, a, nbricks, nullptr, nullptr, st) : launch_gather<1, true, false>(g, a, nbricks, nullptr, nullptr, st);
      default:
        return one_round ? launch_gather<9, true, true>(g, a, nbricks, nullptr, nullptr, st) : launch_gather<9, true, false>(g, a, nbricks, nullptr, nullptr, st);
    }
  }
  if (slab4) {  // 4 x 8 x 8 bricks into gradient tensors: SH degree 0 / 2 (the lists of a deferred gradient bucket, summed on demand)
    if (K != 1 && K != 9) return RF_ERR_UNSUPPORTED;
    return K == 1 ? launch_gather<1, false, true, false, 4>(g, a, nbricks, grad_densities_dev, grad_features_dev, st)
                  : launch_gather<9, false, true, false, 4>(g, a, nbricks, grad_densities_dev, grad_features_dev, st);
  }
  switch (K) {
    case 1:
      return shift == 3 ? launch_gather<1, false, true>(g, a, nbricks, grad_densities_dev, grad_features_dev, st)
                        : launch_gather<1, false, false>(g, a, nbricks, grad_densities_dev, grad_features_dev, st);
    case 4:
      return shift == 3 ? launch_gather<4, false, true>(g, a, nbricks, grad_densities_dev, grad_features_dev, st)
                        : launch_gather<4, false, false>(g, a, nbricks, grad_densities_dev, grad_features_dev, st);
    case 9:
      return shift == 3 ? launch_gather<9, false, true>(g, a, nbricks, grad_densities_dev, grad_features_dev, st)
                        : launch_gather<9, false, false>(g, a, nbricks, grad_densities_dev, grad_features_dev, st);
    default:
      return shift == 3 ? launch_gather<16, false, true>(g, a, nbricks, grad_densities_dev, grad_features_dev, st)
                        : launch_gather<16, false, false>(g, a, nbricks, grad_densities_dev, grad_features_dev, st);
  }
}

int rf_brick_accumulate(const RFGrid* grid, int32_t brick_size, const RFBrickList* lists, int32_t num_lists,
                        float* grad_densities_dev, float* grad_features_dev, int32_t accumulate, void* stream) {
  return brick_accumulate_impl(grid, brick_size, lists, num_lists, grad_densities_dev, grad_features_dev, accumulate, nullptr, 0, 0, stream);
}

int rf_brick_accumulate_adam(const RFGrid* grid, int32_t brick_size, const RFBrickList* lists, int32_t num_lists,
                             const RFAdamState* adam, void* stream) {
  if (!adam) return RF_ERR_NULL_POINTER;
  return brick_accumulate_impl(grid, brick_size, lists, num_lists, nullptr, nullptr, 0, adam, 0, 0, stream);
}

int rf_brick_accumulate_adam_mirror(const RFGrid* grid, int32_t brick_size, const RFBrickList* lists, int32_t num_lists,
                                    const RFAdamState* adam, float* mirror_densities_dev, float* mirror_features_dev, void* stream) {
  if (!adam || !mirror_densities_dev || !mirror_features_dev) return RF_ERR_NULL_POINTER;
  if (brick_size != RF_BRICK_4X8X8) return RF_ERR_UNSUPPORTED;
  return brick_accumulate_impl(grid, brick_size, lists, num_lists, mirror_densities_dev, mirror_features_dev, 0, adam, 0, 0, stream);
}

int rf_brick_accumulate_adam_range(const RFGrid* grid, int32_t brick_size, const RFBrickList* lists, int32_t num_lists,
                                   const RFAdamState* adam, int32_t first_brick, int32_t num_bricks, void* stream) {
  if (!adam) return RF_ERR_NULL_POINTER;
  if (num_bricks < 1) return num_bricks == 0 ? RF_OK : RF_ERR_BAD_SHAPE;
  return brick_accumulate_impl(grid, brick_size, lists, num_lists, nullptr, nullptr, 0, adam, first_brick, num_bricks, stream);
}

int64_t rf_brick_split_scratch_bytes(const RFGrid* grid, int32_t num_bricks, int32_t parts) {
  if (!grid || num_bricks < 1 || parts < 1 || parts > kMaxListsPerKind) return -1;
  const int K = grid->num_features / 3;
  const long long words = brick_acc_words(8, 3 * K + 1);
  return ((long long)num_bricks * (1 + parts) * 4 + 255) / 256 * 256 + (long long)num_bricks * parts * words * 4;
}

int rf_brick_accumulate_adam_split(const RFGrid* grid, int32_t brick_size, const RFBrickList* lists, int32_t num_lists,
                                   const RFAdamState* adam, int32_t first_brick, int32_t num_bricks, int32_t parts, void* scratch_dev,
                                   int64_t scratch_bytes, void* stream) {
  if (!adam) return RF_ERR_NULL_POINTER;
  if (num_bricks < 1) return num_bricks == 0 ? RF_OK : RF_ERR_BAD_SHAPE;
  if (parts < 1) return RF_ERR_BAD_SHAPE;
  return brick_accumulate_impl(grid, brick_size, lists, num_lists, nullptr, nullptr, 0, adam, first_brick, num_bricks, stream, parts, scratch_dev, scratch_bytes);
}

int rf_grid_query(const RFGrid* grid, const float* points_dev, int64_t num_points, float* out_dev, void* stream) {
  const int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  if (num_points == 0) return RF_OK;
  if (!points_dev || !out_dev) return RF_ERR_NULL_POINTER;
  if (num_points < 0) return RF_ERR_BAD_SHAPE;
  const GridArgs g = to_args(grid);
  const long long total = (long long)num_points * (g.F + 1);
  hipLaunchKernelGGL(grid_query_kernel, dim3(grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, g, points_dev,
                     (long long)num_points, out_dev);
  return launch_status();
}

int rf_grid_query_backward(const RFGrid* grid, const float* points_dev, int64_t num_points, const float* grad_out_dev,
                           float* grad_densities_dev, float* grad_features_dev, void* stream) {
  const int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  if (num_points == 0) return RF_OK;
  if (!points_dev || !grad_out_dev || !grad_densities_dev) return RF_ERR_NULL_POINTER;
  if (!grad_features_dev && !(grid->layout != RF_LAYOUT_REFERENCE && grid->num_features == 3)) return RF_ERR_NULL_POINTER;
  if (num_points < 0) return RF_ERR_BAD_SHAPE;
  const GridArgs g = to_args(grid);
  const long long total = (long long)num_points * (g.F + 1);
  hipLaunchKernelGGL(grid_query_backward_kernel, dim3(grid_1d(total, 256)), dim3(256), 0, (hipStream_t)stream, g,
                     points_dev, (long long)num_points, grad_out_dev, grad_densities_dev, grad_features_dev);
  return launch_status();
}

int rf_build_occupancy(const RFGrid* grid, float threshold, uint32_t* occupancy_dev, void* stream) {
  const int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  if (!occupancy_dev) return RF_ERR_NULL_POINTER;
  const GridArgs g = to_args(grid);
  const long long ncell = (long long)(g.X + 1) * (g.Y + 1) * (g.Z + 1);
  const long long nwords = (ncell + 31) / 32;
  hipLaunchKernelGGL(build_occupancy_kernel, dim3(grid_1d(ncell, 256, 256LL * 32)), dim3(256), 0, (hipStream_t)stream, g,
                     threshold, occupancy_dev, nwords);
  return launch_status();
}

int rf_upsample_grid(const RFGrid* src, const RFGrid* dst, void* stream) {
  int rc = check_grid(src);
  if (rc != RF_OK) return rc;
  rc = check_grid(dst);
  if (rc != RF_OK) return rc;
  if (src->num_features != dst->num_features) return RF_ERR_BAD_SHAPE;
  if (src->densities_dev == dst->densities_dev || (dst->features_dev && src->features_dev == dst->features_dev)) return RF_ERR_BAD_SHAPE;
  const GridArgs gs = to_args(src), gd = to_args(dst);
  const long long total = (long long)gd.X * gd.Y * gd.Z * (gd.F + 1);
  hipLaunchKernelGGL(upsample_grid_kernel, dim3(grid_1d(total, 256, 256LL * 64)), dim3(256), 0, (hipStream_t)stream, gs, gd,
                     const_cast<float*>(dst->densities_dev), const_cast<float*>(dst->features_dev), total);
  return launch_status();
}

int rf_convert_grid(const RFGrid* src, const RFGrid* dst, void* stream) {
  int rc = check_grid(src);
  if (rc != RF_OK) return rc;
  rc = check_grid(dst);
  if (rc != RF_OK) return rc;
  if (src->num_features != dst->num_features) return RF_ERR_BAD_SHAPE;
  for (int a = 0; a < 3; ++a)
    if (src->dims[a] != dst->dims[a]) return RF_ERR_BAD_SHAPE;
  if (src->densities_dev == dst->densities_dev || (dst->features_dev && src->features_dev == dst->features_dev)) return RF_ERR_BAD_SHAPE;
  const GridArgs gs = to_args(src), gd = to_args(dst);
  const unsigned int nodes = (unsigned)gd.X * (unsigned)gd.Y * (unsigned)gd.Z;
  const int K = gd.F / 3;
  if (src->layout == RF_LAYOUT_REFERENCE && dst->layout == RF_LAYOUT_SPLIT && (K == 1 || K == 9) && dst->density_stride == 4 &&
      (K == 1 || dst->feature_stride == gd.F - 3) && (((uintptr_t)dst->densities_dev | (uintptr_t)dst->features_dev) & 15u) == 0 &&
      (unsigned long long)nodes * 7ull < (1ull << 32)) {
    float4* base = reinterpret_cast<float4*>(const_cast<float*>(dst->densities_dev));
    float4* rest = reinterpret_cast<float4*>(const_cast<float*>(dst->features_dev));
    const unsigned blocks = grid_1d((long long)nodes * ((gd.F + 1) / 4), 256, 256LL * 64);
    if (K == 1)
      hipLaunchKernelGGL((reference_to_split_kernel<1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, gs.dens, gs.feat, gs.dstride, gs.fstride, base, rest, nodes);
    else
      hipLaunchKernelGGL((reference_to_split_kernel<9>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, gs.dens, gs.feat, gs.dstride, gs.fstride, base, rest, nodes);
    return launch_status();
  }
  if (src->layout == RF_LAYOUT_SPLIT && dst->layout == RF_LAYOUT_REFERENCE && (K == 1 || K == 9) && src->density_stride == 4 &&
      (K == 1 || src->feature_stride == gd.F - 3) && (((uintptr_t)src->densities_dev | (uintptr_t)src->features_dev) & 15u) == 0 &&
      (unsigned long long)nodes * 7ull < (1ull << 32)) {
    const float4* base = reinterpret_cast<const float4*>(src->densities_dev);
    const float4* rest = reinterpret_cast<const float4*>(src->features_dev);
    float* dd = const_cast<float*>(dst->densities_dev);
    float* df = const_cast<float*>(dst->features_dev);
    if (dst->density_stride == 1 && dst->feature_stride == gd.F && (nodes & 3u) == 0 && (((uintptr_t)dd | (uintptr_t)df) & 15u) == 0 &&
        (unsigned long long)nodes * (unsigned long long)gd.F < (1ull << 32)) {  // contiguous reference tensors: 16-byte stores
      const unsigned blocks4 = grid_1d((long long)nodes * gd.F / 4 + nodes / 4, 256, 256LL * 64);
      if (K == 1)
        hipLaunchKernelGGL((split_to_reference_quads_kernel<1>), dim3(blocks4), dim3(256), 0, (hipStream_t)stream, src->densities_dev, src->features_dev,
                           reinterpret_cast<float4*>(dd), reinterpret_cast<float4*>(df), nodes);
      else
        hipLaunchKernelGGL((split_to_reference_quads_kernel<9>), dim3(blocks4), dim3(256), 0, (hipStream_t)stream, src->densities_dev, src->features_dev,
                           reinterpret_cast<float4*>(dd), reinterpret_cast<float4*>(df), nodes);
      return launch_status();
    }
    const unsigned blocks = grid_1d((long long)nodes * ((gd.F + 1) / 4), 256, 256LL * 64);
    if (K == 1)
      hipLaunchKernelGGL((split_to_reference_kernel<1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, base, rest, dd, df, gd.dstride, gd.fstride, nodes);
    else
      hipLaunchKernelGGL((split_to_reference_kernel<9>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, base, rest, dd, df, gd.dstride, gd.fstride, nodes);
    return launch_status();
  }
  const int cw = gd.F + 1 <= 32 ? 32 : 64, per_block = 256 / cw;
  hipLaunchKernelGGL(convert_grid_kernel, dim3(grid_1d(nodes, per_block, 256LL * 256)), dim3(cw, per_block), 0, (hipStream_t)stream, gs, gd,
                     const_cast<float*>(dst->densities_dev), const_cast<float*>(dst->features_dev), nodes);
  return launch_status();
}

static int l1_loss_grad_impl(const L1Sets& sets, int nsets, const float* target_dev, int64_t num_rays, float scale, void* stream) {
  const long long n3 = (long long)num_rays * 3;
  hipLaunchKernelGGL(l1_loss_grad_kernel, dim3(grid_1d(n3, kBlock * 8, 32), nsets), dim3(kBlock), 0, (hipStream_t)stream, sets,
                     target_dev, n3, scale / (float)n3);
  return launch_status();
}

int rf_l1_loss_grad(const float* colour_dev, const float* target_dev, int64_t num_rays, float scale,
                    float* grad_colour_dev, float* sums_dev, void* stream) {
  if (num_rays == 0) return RF_OK;
  if (!colour_dev || !target_dev || !grad_colour_dev || !sums_dev) return RF_ERR_NULL_POINTER;
  if (num_rays < 0) return RF_ERR_BAD_SHAPE;
  L1Sets sets = {};
  sets.colour[0] = colour_dev;
  sets.grad[0] = grad_colour_dev;
  sets.sums[0] = sums_dev;
  return l1_loss_grad_impl(sets, 1, target_dev, num_rays, scale, stream);
}

int rf_l1_loss_grad_pair(const float* const* colour_dev, const float* target_dev, int64_t num_rays, float scale,
                         float* const* grad_colour_dev, float* workspace_dev, float* out_dev, void* stream) {
  if (!colour_dev || !grad_colour_dev || !target_dev || !workspace_dev || !out_dev) return RF_ERR_NULL_POINTER;
  if (num_rays <= 0) return RF_ERR_BAD_SHAPE;
  L1Sets sets = {};
  for (int i = 0; i < 2; ++i) {
    if (!colour_dev[i] || !grad_colour_dev[i]) return RF_ERR_NULL_POINTER;
    sets.colour[i] = colour_dev[i];
    sets.grad[i] = grad_colour_dev[i];
    sets.sums[i] = workspace_dev + 2 * i;
  }
  const long long n3 = (long long)num_rays * 3;
  hipLaunchKernelGGL(l1_loss_pair_kernel, dim3(grid_1d(n3, kBlock * 8, 32), 2), dim3(kBlock), 0, (hipStream_t)stream, sets, target_dev, n3,
                     scale / (float)n3, 1.0f / (float)n3, out_dev);
  return launch_status();
}

int rf_adam_step(float* param_dev, float* grad_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t numel,
                 float lr, float beta1, float beta2, float eps, int32_t step, int32_t zero_grad, void* stream) {
  if (numel == 0) return RF_OK;
  if (!param_dev || !grad_dev || !exp_avg_dev || !exp_avg_sq_dev) return RF_ERR_NULL_POINTER;
  if (numel < 0 || step < 1) return RF_ERR_BAD_SHAPE;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const bool aligned = (((uintptr_t)param_dev | (uintptr_t)grad_dev | (uintptr_t)exp_avg_dev | (uintptr_t)exp_avg_sq_dev) & 15u) == 0;
  const dim3 block(256);
  const dim3 grid(aligned ? (unsigned)((numel / 4 + 256) / 256) : grid_1d(numel / 4 + 1, 256, 256LL * 32));
  hipStream_t st = (hipStream_t)stream;
#define RF_ADAM(Z, S)                                                                                                    \
  hipLaunchKernelGGL((adam_kernel<Z, S>), grid, block, 0, st, param_dev, grad_dev, exp_avg_dev, exp_avg_sq_dev,          \
                     (long long)numel, lr, beta1, beta2, eps, (float)bc1, (float)(1.0 / sqrt(bc2)))
  if (zero_grad) {
    if (aligned) RF_ADAM(true, true); else RF_ADAM(true, false);
  } else {
    if (aligned) RF_ADAM(false, true); else RF_ADAM(false, false);
  }
#undef RF_ADAM
  return launch_status();
}

int rf_train_step(const RFGrid* grid, const RFTrainStep* step, void* stream) {
  if (!grid || !step) return RF_ERR_NULL_POINTER;
  if (step->num_rays == 0) return RF_OK;
  if (!step->origins_dev || !step->directions_dev || !step->pixels_dev || !step->loss_sums_dev) return RF_ERR_NULL_POINTER;
  if (step->num_rays < 0) return RF_ERR_BAD_SHAPE;
  for (int i = 0; i < 2; ++i) {
    const RFPassScratch& ps = step->pass[i];
    if (!ps.grad_colour_dev || !ps.cursor_dev || !ps.offsets_dev || !ps.records_sorted_dev || !ps.out.key_hist_dev) return RF_ERR_NULL_POINTER;
  }
  int shift, nb[3];
  int rc = check_grid(grid);
  if (rc != RF_OK) return rc;
  rc = brick_geometry(grid, step->pass[0].out.brick_size, &shift, nb, false);
  if (rc != RF_OK) return rc;
  if (step->pass[1].out.brick_size != step->pass[0].out.brick_size) return RF_ERR_BAD_SHAPE;
  const int nkeys = nb[0] * nb[1] * nb[2] * 8;
  if (nkeys > (1 << 21)) return RF_ERR_BAD_SHAPE;
  const bool run_forward = step->phases == 0 || (step->phases & RF_STEP_FORWARD), run_emit = step->phases == 0 || (step->phases & RF_STEP_EMIT),
             run_bricks = step->phases == 0 || (step->phases & RF_STEP_BRICKS);
  const bool emit_one[2] = {run_emit || (step->phases & RF_STEP_EMIT_SPECULAR) != 0, run_emit || (step->phases & RF_STEP_EMIT_DIFFUSE) != 0};
  // the forward part in two pieces, diffuse render first (it reads the base tensor only: a data-parallel caller lets it run while the
  // all-gather of the `rest` parameters of the previous iteration is still arriving)
  const bool fwd_a = (step->phases & RF_STEP_SELECT_AND_DIFFUSE_FORWARD) != 0, fwd_b = (step->phases & RF_STEP_SPECULAR_FORWARD_AND_LOSSES) != 0;
  if (run_bricks) {  // everything the last launch would refuse is refused before the first one
    if (step->adam) {
      rc = check_fused_adam(grid, step->adam, grid->num_features / 3, false);
      if (rc != RF_OK) return rc;
    } else if (!step->grad_first_dev || (!step->grad_second_dev && !(grid->layout != RF_LAYOUT_REFERENCE && grid->num_features == 3))) {
      return RF_ERR_NULL_POINTER;
    }
    }
  const float loss_scale = step->loss_scale != 0.0f ? step->loss_scale : 1.0f;
  hipStream_t st = (hipStream_t)stream;
  int ev = 0;
#define RF_STEP_EVENT()                                                                                      \
  do {                                                                                                       \
    if (events && hipEventRecord((hipEvent_t)events[ev++], st) != hipSuccess) return RF_ERR_LAUNCH; \
  } while (0)
  void* const* events = (run_forward && run_emit && run_bricks) ? step->timing_events : nullptr;  // (the per-launch events describe the whole iteration)
  RFRayBatch rays[2];
  uint32_t flags[2];
  for (int i = 0; i < 2; ++i) {
    const RFPassScratch& ps = step->pass[i];
    rays[i] = RFRayBatch{};
    rays[i].origins_dev = step->origins_dev;
    rays[i].directions_dev = step->directions_dev;
    rays[i].num_rays = step->num_rays;
    rays[i].num_samples = step->num_samples;
    rays[i].near = step->near;
    rays[i].far = step->far;
    rays[i].t_vals_dev = step->t_vals_dev;
    rays[i].t_rand_dev = ps.t_rand_dev;
    rays[i].jitter_key = ps.jitter_key;
    rays[i].first_ray = step->first_ray;
    flags[i] = (step->flags & ~(uint32_t)RF_FLAG_RENDER_DIFFUSE) | (i == 1 ? (uint32_t)RF_FLAG_RENDER_DIFFUSE : 0u);
  }
  auto launch_select = [&]() -> int {
    if (step->select) {
      const RFRaySelection* s = step->select;
      return select_impl(s->height, s->width, s->focal, s->poses_dev, s->image_ids_dev, s->num_batch_images, s->pixel_table_dev, s->key,
                         s->first_index, step->num_rays, step->origins_dev, step->directions_dev, step->pixels_dev, nullptr,
                         step->loss_sums_dev /* cleared by the same launch */, stream);
    }
    return hipMemsetAsync(step->loss_sums_dev, 0, 4 * sizeof(float), st) == hipSuccess ? RF_OK : RF_ERR_LAUNCH;
  };
  auto launch_losses_and_offsets = [&](int passes) -> int {  // the losses of both renders and the offsets of both record lists in one launch
    L1Sets sets = {};
    BinLists bl = {};
    for (int k = 0; k < 2; ++k) {
      sets.colour[k] = step->pass[k].out.colour_dev;
      sets.grad[k] = step->pass[k].grad_colour_dev;
      sets.sums[k] = step->loss_sums_dev + 2 * k;
      bl.hist[k] = step->pass[k].out.key_hist_dev;
      bl.offsets[k] = reinterpret_cast<long long*>(step->pass[k].offsets_dev);
      bl.cursor[k] = step->pass[k].cursor_dev;
    }
    const long long n3 = (long long)step->num_rays * 3;
    const int loss_blocks = (int)grid_1d(n3, 1024 * 2, 64), offset_blocks = (nkeys + 1023) / 1024;
    hipLaunchKernelGGL(loss_and_offsets_kernel, dim3(loss_blocks > offset_blocks ? loss_blocks : offset_blocks, 4), dim3(1024), 0, st, sets,
                       step->pixels_dev, n3, loss_scale / (float)n3, loss_blocks, bl, nkeys, offset_blocks, passes);
    return launch_status();
  };
  if (run_forward) {
    RF_STEP_EVENT();
    rc = launch_select();
    if (rc != RF_OK) return rc;
    RF_STEP_EVENT();
    static const bool pair_forwards = [] {
      const char* e = getenv("RF_FWD_PAIR");
      return e ? atoi(e) != 0 : true;
    }();
    bool paired = false;
    if (pair_forwards) {  // both renders in one launch: the second finds the base records of its corners on chip
      const RFRayBatch* const rr[2] = {&rays[0], &rays[1]};
      const RFRenderOut* const oo[2] = {&step->pass[0].out, &step->pass[1].out};
      rc = forward_pair_impl(grid, rr, flags, oo, stream);
      if (rc != RF_OK && rc != RF_ERR_UNSUPPORTED) return rc;
      paired = rc == RF_OK;
    }
    for (int i = 0; i < 2; ++i) {
      if (!paired) {
        rc = rf_render_forward(grid, &rays[i], flags[i], &step->pass[i].out, stream);
        if (rc != RF_OK) return rc;
      }
      RF_STEP_EVENT();  // (paired: the first forward slot holds the launch, the second is empty)
      if (i == 1) {
        rc = launch_losses_and_offsets(3);
        if (rc != RF_OK) return rc;
      }
      RF_STEP_EVENT();
    }
  } else {
    if (fwd_a) {
      rc = launch_select();
      if (rc != RF_OK) return rc;
      rc = rf_render_forward(grid, &rays[1], flags[1], &step->pass[1].out, stream);
      if (rc != RF_OK) return rc;
    }
    if (fwd_b) {
      rc = rf_render_forward(grid, &rays[0], flags[0], &step->pass[0].out, stream);
      if (rc != RF_OK) return rc;
      rc = launch_losses_and_offsets(3);
      if (rc != RF_OK) return rc;
    }
    // the two renders' chains apart (pipelined owner-computes step): everything of the render_diffuse pass that reads the base tensor
    // only and needs no other rank -- selection, forward, loss + offsets of ITS list, adjoint -- then the specular forward + its loss/offsets
    if (step->phases & RF_STEP_DIFFUSE_CHAIN) {
      rc = launch_select();
      if (rc != RF_OK) return rc;
      rc = rf_render_forward(grid, &rays[1], flags[1], &step->pass[1].out, stream);
      if (rc != RF_OK) return rc;
      rc = launch_losses_and_offsets(2);
      if (rc != RF_OK) return rc;
      const RFPassScratch& ps = step->pass[1];
      const RFRenderGrads grads = {ps.grad_colour_dev, nullptr, nullptr};
      rc = rf_render_backward_emit_direct(grid, &rays[1], flags[1], &ps.out, &grads, ps.out.brick_size, ps.cursor_dev, ps.records_sorted_dev, ps.out.key_hist_dev, stream);
      if (rc != RF_OK) return rc;
    }
    if (step->phases & RF_STEP_SPECULAR_FORWARD) {
      rc = rf_render_forward(grid, &rays[0], flags[0], &step->pass[0].out, stream);
      if (rc != RF_OK) return rc;
      rc = launch_losses_and_offsets(1);
      if (rc != RF_OK) return rc;
    }
  }
  if (emit_one[0] || emit_one[1]) {
    static const bool pair_emits = [] {
      const char* e = getenv("RF_EMIT_PAIR");
      return e ? atoi(e) != 0 : true;
    }();
    bool paired = false;
    if (pair_emits && emit_one[0] && emit_one[1]) {
      RF_STEP_EVENT();
      const RFRayBatch* const rr[2] = {&rays[0], &rays[1]};
      const RFPassScratch* const pp[2] = {&step->pass[0], &step->pass[1]};
      rc = emit_pair_impl(grid, rr, flags, pp, stream);
      if (rc != RF_OK && rc != RF_ERR_UNSUPPORTED) return rc;
      paired = rc == RF_OK;
      if (paired) {  // (the first emit slot holds the launch, the second is empty)
        RF_STEP_EVENT();
        RF_STEP_EVENT();
        RF_STEP_EVENT();
      } else if (events) {
        --ev;
      }
    }
    for (int i = 0; i < 2 && !paired; ++i) {
      if (!emit_one[i]) continue;
      const RFPassScratch& ps = step->pass[i];
      const RFRenderGrads grads = {ps.grad_colour_dev, nullptr, nullptr};
      RF_STEP_EVENT();  // (offsets[0] = the launch above, offsets[1] = nothing)
      rc = rf_render_backward_emit_direct(grid, &rays[i], flags[i], &ps.out, &grads, ps.out.brick_size, ps.cursor_dev, ps.records_sorted_dev,
                                          ps.out.key_hist_dev, stream);
      if (rc != RF_OK) return rc;
      RF_STEP_EVENT();
    }
  }
  if (run_bricks) {
    RFBrickList lists[2];
    for (int i = 0; i < 2; ++i) lists[i] = RFBrickList{step->pass[i].records_sorted_dev, step->pass[i].offsets_dev, i};
    if (step->adam)
      rc = rf_brick_accumulate_adam(grid, step->pass[0].out.brick_size, lists, 2, step->adam, stream);
    else
      rc = rf_brick_accumulate(grid, step->pass[0].out.brick_size, lists, 2, step->grad_first_dev, step->grad_second_dev, 0, stream);
    if (rc != RF_OK) return rc;
    RF_STEP_EVENT();
  }
#undef RF_STEP_EVENT
  return RF_OK;
}

}  // extern "C"

// Microbenchmark: variants of the fused Adam streaming kernel (4 reads + 3 writes of 4 B per parameter) on 58.7 M
// parameters (128^3 x 28).  hipcc --offload-arch=gfx950 -O3 tools/adam_microbench.hip -o tools/adam_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ntload(const float4* a) { vf4 t = __builtin_nontemporal_load(reinterpret_cast<const vf4*>(a)); return make_float4(t.x, t.y, t.z, t.w); }
__device__ __forceinline__ void ntstore(float4 x, float4* a) { vf4 t = {x.x, x.y, x.z, x.w}; __builtin_nontemporal_store(t, reinterpret_cast<vf4*>(a)); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ float4 upd(float4 g, float4& p, float4& m, float4& v, float lr, float b1, float b2, float eps, float bc1, float bc2s) {
  float* G = &g.x; float* P = &p.x; float* M = &m.x; float* V = &v.x;
  const float step = lr / bc1;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    M[c] = M[c] + (G[c] - M[c]) * (1.0f - b1);
    V[c] = V[c] * b2 + (G[c] * G[c]) * (1.0f - b2);
    P[c] = P[c] - step * (M[c] / (sqrtf(V[c]) / bc2s + eps));
  }
  return p;
}

// MODE 0: plain grid-stride;  1: nontemporal loads of g and nontemporal stores of m, v;  2: all nontemporal;
// 3: two float4 per thread per iteration (8 loads in flight)
template <int MODE>
__global__ __launch_bounds__(256) void k(float4* p, const float4* g, float4* m, float4* v, long long n4) {
  const float lr = 0.03f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f, bc1 = 0.1f, bc2s = 0.0316f;
  const long long stride = (long long)gridDim.x * blockDim.x;
  if (MODE == 3) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
      const long long j = i + stride;
      float4 g0 = g[i], p0 = p[i], m0 = m[i], v0 = v[i];
      float4 g1, p1, m1, v1;
      const bool two = j < n4;
      if (two) { g1 = g[j]; p1 = p[j]; m1 = m[j]; v1 = v[j]; }
      upd(g0, p0, m0, v0, lr, b1, b2, eps, bc1, bc2s);
      p[i] = p0; m[i] = m0; v[i] = v0;
      if (two) { upd(g1, p1, m1, v1, lr, b1, b2, eps, bc1, bc2s); p[j] = p1; m[j] = m1; v[j] = v1; }
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 gg, pp, mm, vv;
    if (MODE >= 1) gg = ntload(&g[i]); else gg = g[i];
    if (MODE == 2) { pp = ntload(&p[i]); mm = ntload(&m[i]); vv = ntload(&v[i]); }
    else { pp = p[i]; mm = m[i]; vv = v[i]; }
    upd(gg, pp, mm, vv, lr, b1, b2, eps, bc1, bc2s);
    if (MODE == 2) ntstore(pp, &p[i]); else p[i] = pp;
    if (MODE >= 1) { ntstore(mm, &m[i]); ntstore(vv, &v[i]); } else { m[i] = mm; v[i] = vv; }
  }
}

template <int MODE>
void run(const char* name, float4* p, float4* g, float4* m, float4* v, long long n4, int iters_per_thread) {
  const int blocks = (int)((n4 + 256LL * iters_per_thread - 1) / (256LL * iters_per_thread));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, p, g, m, v, n4);
  CK(hipDeviceSynchronize());
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, p, g, m, v, n4);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  printf("%-34s iters/thread %3d  blocks %6d  %7.4f ms  %6.2f TB/s\n", name, iters_per_thread, blocks, best, n4 * 16.0 * 7 / (best * 1e-3) / 1e12);
}

int main() {
  const long long n = 128LL * 128 * 128 * 28, n4 = n / 4;
  float4 *p, *g, *m, *v;
  CK(hipMalloc(&p, n * 4)); CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&m, n * 4)); CK(hipMalloc(&v, n * 4));
  CK(hipMemset(p, 0, n * 4)); CK(hipMemset(g, 0, n * 4)); CK(hipMemset(m, 0, n * 4)); CK(hipMemset(v, 0, n * 4));
  for (int it : {1, 4, 32}) {
    run<0>("plain", p, g, m, v, n4, it);
    run<1>("nt load g, nt store m v", p, g, m, v, n4, it);
    run<2>("all nontemporal", p, g, m, v, n4, it);
    run<3>("two float4 per iteration", p, g, m, v, n4, it);
  }
  return 0;
}

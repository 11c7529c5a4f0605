// Microbenchmark 3: LDS accumulate primitives with precomputed addresses (no index arithmetic in the loop).
// hipcc --offload-arch=gfx950 -O3 tools/lds_microbench3.hip -o tools/lds_microbench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NADDR = 64;  // address table entries per lane (in LDS)
// MODE 0: ds_add_f32  1: plain RMW chain  2: ds_add_u32  3: ds_add_f64  4: plain RMW, 2 independent per step  5: plain RMW, 4 independent per step
template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, int iters, int acc_words) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  float* acc = lds;
  int* addr = reinterpret_cast<int*>(lds + acc_words * (MODE == 3 ? 2 : 1));  // [NADDR][64]
  for (int i = lane; i < acc_words * (MODE == 3 ? 2 : 1); i += 64) acc[i] = 0.f;
  const int q = lane >> 3, c = lane & 7;
  const int SY = 8 * 8 + 8, SX = 8 * SY + 16;
  const int corner = ((q >> 2) & 1) * SX + ((q >> 1) & 1) * SY + (q & 1) * 8 + c;
  unsigned s = blockIdx.x * 977u + 7u;
  for (int i = 0; i < NADDR; ++i) {
    s = s * 1664525u + 1013904223u;
    // groups of 4 consecutive entries are node-disjoint: x differs by 2 within the group
    const int x = ((s >> 10) % 3u) * 0 + (i & 3) * 2 - ((i & 3) == 3 ? 1 : 0) * 0, y = (s >> 16) % 7u, z = (s >> 24) % 7u;
    const int xx = (i & 3) * 2 > 6 ? 6 : (i & 3) * 2;
    addr[i * 64 + lane] = (MODE >= 4 ? xx : (int)((s >> 10) % 7u)) * SX + y * SY + z * 8 + corner + x * 0;
  }
  __syncthreads();
  float v = 1.0f + lane;
  for (int it = 0; it < iters; it += 4) {
    const int* ap = addr + ((it & (NADDR - 1)) * 64) + lane;
    const int a0 = ap[0], a1 = ap[64], a2 = ap[128], a3 = ap[192];
    if (MODE == 0) { atomicAdd(&acc[a0], v); atomicAdd(&acc[a1], v); atomicAdd(&acc[a2], v); atomicAdd(&acc[a3], v); }
    else if (MODE == 1) { acc[a0] += v; acc[a1] += v; acc[a2] += v; acc[a3] += v; }
    else if (MODE == 2) { unsigned* u = reinterpret_cast<unsigned*>(acc); atomicAdd(&u[a0], 1u); atomicAdd(&u[a1], 1u); atomicAdd(&u[a2], 1u); atomicAdd(&u[a3], 1u); }
    else if (MODE == 3) { double* d = reinterpret_cast<double*>(acc); atomicAdd(&d[a0], (double)v); atomicAdd(&d[a1], (double)v); atomicAdd(&d[a2], (double)v); atomicAdd(&d[a3], (double)v); }
    else if (MODE == 4) {
      float r0 = acc[a0], r1 = acc[a1]; acc[a0] = r0 + v; acc[a1] = r1 + v;
      float r2 = acc[a2], r3 = acc[a3]; acc[a2] = r2 + v; acc[a3] = r3 + v;
    } else {
      float r0 = acc[a0], r1 = acc[a1], r2 = acc[a2], r3 = acc[a3];
      acc[a0] = r0 + v; acc[a1] = r1 + v; acc[a2] = r2 + v; acc[a3] = r3 + v;
    }
  }
  __syncthreads();
  out[blockIdx.x * 64 + threadIdx.x] = acc[threadIdx.x * 7];
}

template <int MODE>
void run(const char* name, float* out, int waves_per_cu) {
  const int iters = 8192;
  const int SX = 8 * (8 * 8 + 8) + 16;
  const int acc_words = 8 * SX;
  size_t lds = (size_t)acc_words * 4 * (MODE == 3 ? 2 : 1) + NADDR * 64 * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int blocks = 256 * waves_per_cu;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), lds, 0, out, 16, acc_words);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(64), lds, 0, out, iters, acc_words);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  printf("%-34s waves/CU %2d  %8.3f ms  %7.2f lanes/clk/CU   %6.1f clk per wave-accumulate\n", name, waves_per_cu, ms,
         (double)blocks * iters * 64 / (ms * 1e-3) / 256 / 2.4e9, (ms * 1e-3) * 2.4e9 / iters);
}

int main() {
  float* out; CK(hipMalloc(&out, 1 << 22));
  for (int w : {1, 4}) {
    run<1>("plain RMW chain", out, w);
    run<4>("plain RMW 2 independent", out, w);
    run<5>("plain RMW 4 independent", out, w);
    run<2>("ds_add_u32", out, w);
    run<3>("ds_add_f64", out, w);
    run<0>("ds_add_f32", out, w);
  }
  run<1>("plain RMW chain", out, 3);
  run<5>("plain RMW 4 independent", out, 3);
  return 0;
}

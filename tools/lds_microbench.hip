// Microbenchmark: LDS accumulation primitives on gfx950 (quoted in DESIGN.md).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_microbench.hip -o tools/lds_microbench
// Every wave performs ITERS accumulations of 64 lanes into a 57 KB LDS array at pseudo-random node offsets
// (lane = channel within a 32-float node record, two nodes per instruction), the access pattern of a brick accumulator.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>  // 0: ds_add_f32   1: plain read-add-write   2: ds_add_u32   3: plain RMW, 2 independent records per iteration
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float acc[];
  const int nodes = 512;
  for (int i = threadIdx.x; i < nodes * 28; i += blockDim.x) acc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 31, half = lane >> 5;
  unsigned h = hash32(blockIdx.x * 977u + wave);
  float v = 1.0f + lane;
  if (c < 28) {
    for (int it = 0; it < iters; ++it) {
      h = hash32(h + it);
      // each wave owns a disjoint channel group in the real kernel; here: disjoint node ranges per wave so that plain RMW is race-free
      const int node = (wave * 128) + ((h >> 3) & 126) + half;
      if (MODE == 0) atomicAdd(&acc[node * 28 + c], v);
      else if (MODE == 1) acc[node * 28 + c] += v;
      else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned*>(&acc[node * 28 + c]), 1u);
      else {
        const int node2 = (wave * 128) + ((h >> 11) & 126) + half;
        const float a = acc[node * 28 + c], b = acc[node2 * 28 + c];
        if (node == node2) acc[node * 28 + c] = a + 2 * v;
        else { acc[node * 28 + c] = a + v; acc[node2 * 28 + c] = b + v; }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = acc[threadIdx.x * 7];
}

template <int MODE>
void run(const char* name, float* out, double accum_per_iter) {
  const int blocks = 256 * 2, iters = 4096;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 57344, 0, out, 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 57344, 0, out, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double waves = blocks * 4.0, total = waves * iters * accum_per_iter * 56;  // 56 active lanes
  printf("%-42s %8.3f ms  %8.1f G lane-accumulations/s  (%.2f lanes/clk/CU at 2.4 GHz)\n", name, ms, total / ms * 1e-6, total / (ms * 1e-3) / 256 / 2.4e9);
}

int main() {
  float* out; CK(hipMalloc(&out, 1 << 20));
  run<0>("ds_add_f32 (LDS float atomic)", out, 1);
  run<1>("plain read-add-write", out, 1);
  run<2>("ds_add_u32 (LDS integer atomic)", out, 1);
  run<3>("plain RMW, 2 records per iteration", out, 2);
  return 0;
}

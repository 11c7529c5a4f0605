// Microbenchmark 2: LDS accumulation throughput on gfx950 as a function of occupancy, with a 2-instruction index
// update (lds_microbench.hip's hash chain hid the LDS cost).  hipcc --offload-arch=gfx950 -O3 tools/lds_microbench2.hip -o tools/lds_microbench2
// lane = 8 corners x 8 channels of a cell; the cell's lower node walks pseudo-randomly through an 8^3-node brick
// accumulator of `CH` channels per node (bank-padded like brick_accumulate_kernel).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: ds_add_f32  1: plain RMW  2: ds_add_u32  3: ds_add_u64 (fixed point)  4: ds_add_rtn-free f32 via 2 halves
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, int iters, int words_per_wave) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* acc = lds + wave * words_per_wave * (MODE == 3 ? 2 : 1);  // every wave its own accumulator: race-free plain RMW
  for (int i = lane; i < words_per_wave * (MODE == 3 ? 2 : 1); i += 64) acc[i] = 0.f;
  const int q = lane >> 3, c = lane & 7;
  const int dx = q >> 2, dy = (q >> 1) & 1, dz = q & 1;
  // 7x7x7 lower nodes; node stride 8 channels (+ row pads 8 / 16 words)
  const int SY = 8 * 8 + 8, SX = 8 * SY + 16;
  const int corner = dx * SX + dy * SY + dz * 8 + c;
  unsigned s = blockIdx.x * 977u + wave * 131u + 7u;
  float v = 1.0f + lane;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const int x = (s >> 10) % 7u, y = (s >> 16) % 7u, z = (s >> 24) % 7u;
    const int a = x * SX + y * SY + z * 8 + corner;
    if (MODE == 0) atomicAdd(&acc[a], v);
    else if (MODE == 1) acc[a] += v;
    else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned*>(&acc[a]), 1u);
    else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long*>(acc) + a, (unsigned long long)(long long)(v * 1048576.f));
  }
  __syncthreads();
  if (threadIdx.x < 64) out[blockIdx.x * 64 + threadIdx.x] = acc[threadIdx.x * 7];
}

template <int MODE, int WAVES>
void run(const char* name, float* out, int wgs_per_cu) {
  const int iters = 8192;
  const int SX = 8 * (8 * 8 + 8) + 16;
  const int words_per_wave = 8 * SX;  // 8^3 nodes x 8 channels, padded
  size_t lds = (size_t)WAVES * words_per_wave * 4 * (MODE == 3 ? 2 : 1);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int blocks = 256 * wgs_per_cu;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), lds, 0, out, 16, words_per_wave);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(WAVES * 64), lds, 0, out, iters, words_per_wave);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double waves = (double)blocks * WAVES, total = waves * iters * 64;
  printf("%-28s waves/CU %2d  LDS/WG %6zu  %8.3f ms  %7.2f lanes/clk/CU   %6.1f clk per wave-iteration\n", name, wgs_per_cu * WAVES, lds, ms,
         total / (ms * 1e-3) / 256 / 2.4e9, (ms * 1e-3) * 2.4e9 / iters);
}

int main() {
  float* out; CK(hipMalloc(&out, 1 << 22));
  run<1, 1>("plain RMW", out, 1);
  run<1, 1>("plain RMW", out, 4);
  run<1, 1>("plain RMW", out, 8);
  run<1, 2>("plain RMW", out, 4);
  run<1, 4>("plain RMW", out, 2);
  run<0, 1>("ds_add_f32", out, 8);
  run<2, 1>("ds_add_u32", out, 1);
  run<2, 1>("ds_add_u32", out, 4);
  run<2, 1>("ds_add_u32", out, 8);
  run<3, 1>("ds_add_u64", out, 1);
  run<3, 1>("ds_add_u64", out, 4);
  return 0;
}

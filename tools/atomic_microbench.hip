// Microbenchmark: float32 global atomic-add throughput on gfx950 as a function of the lane->address pattern.
// Used to choose the scatter layout of the render backward kernel (results quoted in DESIGN.md).
//   build: hipcc --offload-arch=gfx950 -O3 tools/atomic_microbench.hip -o /tmp/atomic_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ inline unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// Each wave performs `iters` rounds; in each round it picks pseudo-random voxel bases and issues atomics with pattern P.
// nvox voxels of `stride` floats.  Patterns:
//  0: 8 groups x 7 lanes, lane l adds 4 dwords (4 instructions) at voxel*stride + 4l + j     (current backward layout)
//  1: 2 groups x 32 lanes (28 active), lane c adds 1 dword at voxel*stride + c                (lane = channel)
//  2: 64 lanes -> 64 different random voxels, 1 dword each                                    (fully scattered)
//  3: like 1 but 8 corners of a cell: voxel, +1, +Z, +Z+1, +YZ ... (realistic 8-corner footprint), 8 instructions
//  4: like 0 but 8 corners (32 instructions)
template <int P>
__global__ void k(float* buf, long long nvox, int stride, int iters, int Z, int YZ) {
  const int lane = threadIdx.x & 63;
  const unsigned wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  for (int it = 0; it < iters; ++it) {
    if (P == 0 || P == 4) {
      const int grp = lane >> 3, sub = lane & 7;
      const long long v = hash32(wid * 9781u + it * 77u + grp) % (nvox - YZ - Z - 2);
      if (sub < 7) {
        const int ncorn = (P == 4) ? 8 : 1;
        for (int c = 0; c < ncorn; ++c) {
          const long long vv = v + (c & 1) + ((c >> 1) & 1) * Z + (c >> 2) * YZ;
          float* p = buf + vv * stride + (sub == 6 ? 23 : 4 * sub);
#pragma unroll
          for (int j = 0; j < 4; ++j) unsafeAtomicAdd(p + j, 1.0f);
        }
      }
    } else if (P == 1 || P == 3) {
      const int grp = lane >> 5, c = lane & 31;
      const long long v = hash32(wid * 9781u + it * 77u + grp) % (nvox - YZ - Z - 2);
      if (c < 28) {
        const int ncorn = (P == 3) ? 8 : 1;
        for (int q = 0; q < ncorn; ++q) {
          const long long vv = v + (q & 1) + ((q >> 1) & 1) * Z + (q >> 2) * YZ;
          unsafeAtomicAdd(buf + vv * stride + c, 1.0f);
        }
      }
    } else {
      const long long v = hash32(wid * 9781u + it * 77u + lane * 131u) % nvox;
      unsafeAtomicAdd(buf + v * stride, 1.0f);
    }
  }
}

template <int P>
void run(const char* name, float* buf, long long nvox, int stride, int Z, int YZ, double lane_atomics_per_wave_iter) {
  const int blocks = 256 * 16, threads = 256, iters = 64;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(threads), 0, 0, buf, nvox, stride, 4, Z, YZ);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(threads), 0, 0, buf, nvox, stride, iters, Z, YZ);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double waves = (double)blocks * threads / 64, total = waves * iters * lane_atomics_per_wave_iter;
  printf("%-44s stride %2d : %8.3f ms  %8.2f G lane-atomics/s  %8.1f GB/s payload\n", name, stride, ms, total / ms * 1e-6, total * 4 / ms * 1e-6);
}

int main() {
  const int G = 128; const long long nvox = (long long)G * G * G;
  for (int stride : {27, 32}) {
    float* buf; CK(hipMalloc(&buf, nvox * stride * sizeof(float))); CK(hipMemset(buf, 0, nvox * stride * sizeof(float)));
    run<0>("P0 8x7 lanes x4 dwords, 1 voxel", buf, nvox, stride, G, G * G, 8 * 7 * 4);
    run<1>("P1 2x28 lanes x1 dword, 1 voxel", buf, nvox, stride, G, G * G, 2 * 28);
    run<2>("P2 64 lanes scattered", buf, nvox, stride, G, G * G, 64);
    run<3>("P3 2x28 lanes, 8 corners", buf, nvox, stride, G, G * G, 2 * 28 * 8);
    run<4>("P4 8x7 lanes x4 dwords, 8 corners", buf, nvox, stride, G, G * G, 8 * 7 * 4 * 8);
    CK(hipFree(buf));
  }
  return 0;
}

// Microbenchmark 4: candidate brick accumulation with LDS float64 atomics (ds_add_f64), every wave of the workgroup
// working on its own record -- no ownership, no tables, no conflict handling.  One workgroup per 8^3-node brick,
// RECS records of 128 B (index quad + 28 channel values) per brick read from global memory.
// hipcc --offload-arch=gfx950 -O3 tools/lds_microbench4.hip -o tools/lds_microbench4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int C = 28;

// MODE 0: ds_add_f64, lane = (corner, channel of 7), 4 instructions per record
// MODE 1: ds_add_f64, lane = (corner, channel of 8), 4 instructions per record (last one half idle)
// MODE 2: the same lanes with ds_add_u64 (fixed point)
// MODE 3: ds_add_f32 (reference point)
// MODE 4: like 0 but 4-channel "diffuse" records only: lane = (record of 2, corner, channel of 4), one instruction per 2 records
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const float4* __restrict__ rec, int recs, float* out, int CS, int SY, int SX) {
  extern __shared__ double acc[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int words = 8 * SX;
  for (int i = tid; i < words; i += WAVES * 64) acc[i] = 0.0;
  __syncthreads();
  const float4* mine = rec + (long long)blockIdx.x * recs * 8;
  if (MODE == 4) {
    const int r2 = lane >> 5, q = (lane >> 2) & 7, c = lane & 3;
    const int dx = q >> 2, dy = (q >> 1) & 1, dz = q & 1;
    for (int r = wave * 2 + r2; r < recs; r += WAVES * 2) {
      const float4 idx = mine[r * 8];
      const float g = reinterpret_cast<const float*>(mine + r * 8 + 1)[c];
      const float fx = floorf(idx.x), fy = floorf(idx.y), fz = floorf(idx.z);
      const float wx = dx ? idx.x - fx : (fx + 1.f) - idx.x, wy = dy ? idx.y - fy : (fy + 1.f) - idx.y, wz = dz ? idx.z - fz : (fz + 1.f) - idx.z;
      const float w = (wx * wy) * wz;
      const int a = ((int)fx + dx) * SX + ((int)fy + dy) * SY + ((int)fz + dz) * CS + c;
      atomicAdd(&acc[a], (double)(w * g));
    }
  } else {
    constexpr int CPL = (MODE == 1 || MODE == 2) ? 8 : 7;
    const int q = lane / CPL >= 8 ? 7 : lane / CPL, c = lane % CPL;
    const bool on = lane < 8 * CPL;
    const int dx = q >> 2, dy = (q >> 1) & 1, dz = q & 1;
    for (int r = wave; r < recs; r += WAVES) {
      const float4 idx = mine[r * 8];
      const float* gv = reinterpret_cast<const float*>(mine + r * 8 + 1);
      float g[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = (c + CPL * j < C) ? gv[c + CPL * j] : 0.f;
      const float fx = floorf(idx.x), fy = floorf(idx.y), fz = floorf(idx.z);
      const float wx = dx ? idx.x - fx : (fx + 1.f) - idx.x, wy = dy ? idx.y - fy : (fy + 1.f) - idx.y, wz = dz ? idx.z - fz : (fz + 1.f) - idx.z;
      const float w = (wx * wy) * wz;
      const int a = ((int)fx + dx) * SX + ((int)fy + dy) * SY + ((int)fz + dz) * CS + c;
      if (on) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (c + CPL * j < C) {
            if (MODE == 2)
              atomicAdd(reinterpret_cast<unsigned long long*>(acc) + a + CPL * j, (unsigned long long)(long long)(w * g[j] * 1099511627776.f));
            else if (MODE == 3)
              atomicAdd(reinterpret_cast<float*>(acc) + a + CPL * j, w * g[j]);
            else
              atomicAdd(&acc[a + CPL * j], (double)(w * g[j]));
          }
        }
      }
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int i = tid; i < words; i += WAVES * 64) s += (float)acc[i];
  out[blockIdx.x * WAVES * 64 + tid] = s;
}

template <int MODE, int WAVES>
void run(const char* name, const float4* rec, int recs, float* out, int bricks, int pad_y, int pad_x) {
  const int CS = (MODE == 4) ? 4 : C;
  const int SY = 8 * CS + pad_y, SX = 8 * SY + pad_x;
  size_t lds = (size_t)8 * SX * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<MODE, WAVES>), dim3(bricks), dim3(WAVES * 64), lds, 0, rec, recs, out, CS, SY, SX);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, WAVES>), dim3(bricks), dim3(WAVES * 64), lds, 0, rec, recs, out, CS, SY, SX);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  ms /= 5;
  const double visits = (double)bricks * recs;
  printf("%-44s waves %2d pad %2d/%2d LDS %6zu  %7.3f ms  %6.1f clk per record per CU  (%.2f M visits)\n", name, WAVES, pad_y, pad_x, lds, ms,
         ms * 1e-3 * 2.4e9 * 256 / visits, visits / 1e6);
}

int main() {
  const int bricks = 4096, recs = 384;
  std::vector<float> h((size_t)bricks * recs * 32);
  unsigned s = 12345u;
  for (size_t r = 0; r < (size_t)bricks * recs; ++r) {
    float* p = &h[r * 32];
    for (int a = 0; a < 3; ++a) { s = s * 1664525u + 1013904223u; p[a] = (float)((s >> 8) % 7000u) / 1000.0f; }  // lower node 0..6
    p[3] = 0.f;
    for (int c = 0; c < 28; ++c) { s = s * 1664525u + 1013904223u; p[4 + c] = (float)(s >> 8) * 1e-9f; }
  }
  float4* rec; CK(hipMalloc(&rec, h.size() * 4));
  CK(hipMemcpy(rec, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  float* out; CK(hipMalloc(&out, (size_t)bricks * 1024 * 4));
  for (int pad : {0, 2, 4}) {
    run<0, 8>("ds_add_f64 8x7 lanes", rec, recs, out, bricks, pad, 0);
    run<0, 16>("ds_add_f64 8x7 lanes", rec, recs, out, bricks, pad, 0);
  }
  run<0, 4>("ds_add_f64 8x7 lanes", rec, recs, out, bricks, 0, 0);
  run<1, 8>("ds_add_f64 8x8 lanes", rec, recs, out, bricks, 0, 0);
  run<1, 16>("ds_add_f64 8x8 lanes", rec, recs, out, bricks, 0, 0);
  run<2, 16>("ds_add_u64 8x8 lanes", rec, recs, out, bricks, 0, 0);
  run<3, 16>("ds_add_f32 8x7 lanes", rec, recs, out, bricks, 0, 0);
  run<4, 4>("ds_add_f64 diffuse 2 records/instr", rec, recs, out, bricks, 0, 0);
  run<4, 8>("ds_add_f64 diffuse 2 records/instr", rec, recs, out, bricks, 0, 0);
  run<4, 16>("ds_add_f64 diffuse 2 records/instr", rec, recs, out, bricks, 0, 0);
  return 0;
}

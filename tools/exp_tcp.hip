// Micro-benchmark (development tool, not part of the product): how the vector L1 (TA/TCP) of a gfx950 CU prices gather patterns.
// Every pattern issues the same number of load instructions per wave; what differs is how the 64 lane addresses of one instruction
// fall into 128-byte lines.  Prints core cycles per load instruction per CU (all CUs busy, 16 waves per CU).
//   hipcc --offload-arch=gfx950 -O3 tools/exp_tcp.hip -o tools/exp_tcp.bin && tools/exp_tcp.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int kIters = 256, kUnroll = 8;

template <int WIDTH>  // dwords per lane
__global__ __launch_bounds__(256) void gather(const char* base, const uint32_t* lane_off, uint32_t iter_stride, uint32_t wrap_mask, float* out, long long* cyc) {
  const int lane = threadIdx.x & 63;
  const uint32_t lo = lane_off[lane] + (threadIdx.x >> 6) * 256u * 0u;
  float acc = 0.f;
  const long long t0 = wall_clock64();
  const long long c0 = __builtin_readcyclecounter();
  uint32_t off = 0;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const uint32_t o = ((off + u * iter_stride) & wrap_mask) + lo;
      if (WIDTH == 1) {
        acc += *reinterpret_cast<const float*>(base + o);
      } else {
        const float4 v = *reinterpret_cast<const float4*>(base + o);
        acc += v.x + v.y + v.z + v.w;
      }
    }
    off += kUnroll * iter_stride;
  }
  const long long c1 = __builtin_readcyclecounter();
  const long long t1 = wall_clock64();
  if (acc == 12345.678f) out[0] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    cyc[0] = c1 - c0;
    cyc[1] = t1 - t0;
  }
}

struct Pattern {
  const char* name;
  int width;
  uint32_t (*off)(int lane);
};

int main() {
  const size_t bytes = 64u << 20;
  char* buf;
  hipMalloc(&buf, bytes);
  hipMemset(buf, 0, bytes);
  uint32_t* d_off;
  hipMalloc(&d_off, 64 * 4);
  float* d_out;
  hipMalloc(&d_out, 4);
  long long* d_cyc;
  hipMalloc(&d_cyc, 16);
  Pattern pats[] = {
      {"dword  64 distinct lines (lane*128)", 1, [](int l) { return (uint32_t)l * 128u; }},
      {"dword  lanes 2i,2i+1 share a line", 1, [](int l) { return (uint32_t)(l >> 1) * 128u + (l & 1) * 16u; }},
      {"dword  lanes 4i..4i+3 share a line", 1, [](int l) { return (uint32_t)(l >> 2) * 128u + (l & 3) * 16u; }},
      {"dword  lanes i,i+32 share a line", 1, [](int l) { return (uint32_t)(l & 31) * 128u + (l >> 5) * 16u; }},
      {"dword  lanes i,i+8 share a line (8 groups interleaved)", 1, [](int l) { return (uint32_t)((l & 7) + (l >> 4) * 8) * 128u + ((l >> 3) & 1) * 16u; }},
      {"dword  all 64 lanes one line, distinct dwords (32 dwords x2)", 1, [](int l) { return (uint32_t)(l & 31) * 4u; }},
      {"dword  fully coalesced 256 B (lane*4)", 1, [](int l) { return (uint32_t)l * 4u; }},
      {"dwordx4 64 distinct lines", 4, [](int l) { return (uint32_t)l * 128u; }},
      {"dwordx4 lanes 2i,2i+1 share a line", 4, [](int l) { return (uint32_t)(l >> 1) * 128u + (l & 1) * 16u; }},
      {"dwordx4 8 lanes per line, contiguous (lane*16)", 4, [](int l) { return (uint32_t)l * 16u; }},
      {"dwordx4 8-lane groups: lane0 own line, lanes1-6 96 B straddling 2 lines (render P1 shape)", 4,
       [](int l) { const int g = l >> 3, s = l & 7; return s == 0 ? (uint32_t)g * 128u : (s == 7 ? (uint32_t)g * 128u : 2048u + (uint32_t)g * 512u + 96u + (uint32_t)(s - 1) * 16u); }},
      {"dwordx4 8-lane groups: lanes1-6 96 B inside ONE line, lanes 0/7 idle-equivalent (same line)", 4,
       [](int l) { const int g = l >> 3, s = l & 7; return (uint32_t)g * 512u + (uint32_t)(s == 0 ? 0 : (s == 7 ? 5 : s - 1)) * 16u; }},
  };
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  printf("CUs %d, clock %d kHz\n", cus, prop.clockRate);
  for (const Pattern& p : pats) {
    uint32_t h_off[64];
    for (int l = 0; l < 64; ++l) h_off[l] = p.off(l);
    hipMemcpy(d_off, h_off, sizeof(h_off), hipMemcpyHostToDevice);
    for (uint32_t wrap : {8u << 10, 4u << 20}) {  // working set per wave: L1-resident (8 KB + pattern extent) or L2-resident
      const uint32_t stride = 8192u + 128u;       // successive instructions move on by this many bytes (different lines)
      const int blocks = cus * 4;                 // 4 blocks x 4 waves = 16 waves per CU
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (p.width == 1)
          gather<1><<<blocks, 256>>>(buf, d_off, stride, wrap - 1, d_out, d_cyc);
        else
          gather<4><<<blocks, 256>>>(buf, d_off, stride, wrap - 1, d_out, d_cyc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long cyc[2];
      hipMemcpy(cyc, d_cyc, 16, hipMemcpyDeviceToHost);
      const double instr_per_cu = 16.0 * kIters * kUnroll;
      printf("%-95s wrap %7u B: %.3f ms, wave-0 cycles/instr-per-CU %.1f  (event-time x 2.4 GHz: %.1f)\n", p.name, wrap, ms, (double)cyc[0] / instr_per_cu,
             ms * 1e-3 * 2.4e9 / instr_per_cu);
    }
  }
  return 0;
}

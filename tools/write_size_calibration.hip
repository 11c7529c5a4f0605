// WRITE_SIZE (rocprofv3 --pmc) against KNOWN byte counts in the store patterns of the adjoint's record lists -- MI355X_MICROARCH.md, HBM:
// "calibrate on a known byte count in your own access pattern before trusting an absolute".  Each kernel below writes exactly
// N_RECORDS records of 48 or 32 bytes (as 16-byte quads, lanes = quads, like render_emit_direct_*), or one wide stream, once:
//
//   stream            16 B per lane, consecutive (the calibrated case of the guide for reads)
//   rec48_runs<L>     48-byte records in runs of L consecutive records at scattered run positions (a key class's run under the
//                     run-aggregated cursor: L = the samples of one ray in one brick)
//   rec32_runs<L>     32-byte records, same
//   ..._nt            the same stores with the non-temporal hint (what the emit kernel uses for its records)
//
//   hipcc --offload-arch=gfx950 -O3 tools/write_size_calibration.hip -o tools/write_size_calibration
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- tools/write_size_calibration     (tools/write_size_calibration.sh)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef float vf4 __attribute__((ext_vector_type(4)));
constexpr long long kRecords = 1 << 21;  // 2 Mi records: 96 MiB of 48-byte records, 64 MiB of 32-byte ones

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}
// a bijection of [0, 2^bits): position of run `i` among the runs
__device__ __forceinline__ uint32_t scatter(uint32_t i, int bits) {
  const uint32_t mask = (1u << bits) - 1u;
  uint32_t x = i;
  x = (x * 0x9E3779B1u) & mask;  // odd multiplier: a permutation of the low `bits` bits
  x ^= x >> (bits / 2);
  x = (x * 0x85EBCA6Bu) & mask;
  return x;
}

template <bool NT>
__global__ void stream(vf4* out, long long quads) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (long long)gridDim.x * blockDim.x) {
    const vf4 v = {(float)q, 1.f, 2.f, 3.f};
    if (NT)
      __builtin_nontemporal_store(v, out + q);
    else
      out[q] = v;
  }
}

// Q quads per record; runs of L records; lane = one quad of one record
template <int Q, int L, bool NT>
__global__ void records(vf4* out, long long nrec, int run_bits) {
  const long long quads = nrec * Q;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < quads; t += (long long)gridDim.x * blockDim.x) {
    const long long rec = t / Q;
    const int quad = (int)(t - rec * Q);
    const long long run = rec / L;
    const long long pos = (long long)scatter((uint32_t)run, run_bits) * L + (rec - run * L);
    const vf4 v = {(float)rec, (float)quad, 2.f, 3.f};
    if (NT)
      __builtin_nontemporal_store(v, out + pos * Q + quad);
    else
      out[pos * Q + quad] = v;
  }
}

static int ilog2(long long x) {
  int b = 0;
  while ((1ll << b) < x) ++b;
  return b;
}

#define CHECK(x)                                                          \
  do {                                                                    \
    hipError_t e_ = (x);                                                  \
    if (e_ != hipSuccess) {                                               \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));             \
      return 1;                                                           \
    }                                                                     \
  } while (0)

template <int Q, int L, bool NT>
static void launch_records(vf4* buf) {
  hipLaunchKernelGGL((records<Q, L, NT>), dim3(4096), dim3(256), 0, 0, buf, kRecords, ilog2(kRecords / L));
}

int main() {
  vf4* buf = nullptr;
  const size_t bytes = (size_t)kRecords * 48;
  CHECK(hipMalloc(&buf, bytes));
  CHECK(hipMemset(buf, 0, bytes));
  CHECK(hipDeviceSynchronize());
  printf("kernel,known_bytes\n");
  hipLaunchKernelGGL((stream<false>), dim3(4096), dim3(256), 0, 0, buf, (long long)(bytes / 16));
  printf("stream<false>,%zu\n", bytes);
  hipLaunchKernelGGL((stream<true>), dim3(4096), dim3(256), 0, 0, buf, (long long)(bytes / 16));
  printf("stream<true>,%zu\n", bytes);
#define REC(Q, L)                                                   \
  launch_records<Q, L, false>(buf);                                 \
  printf("records<%d, %d, false>,%lld\n", Q, L, kRecords * Q * 16); \
  launch_records<Q, L, true>(buf);                                  \
  printf("records<%d, %d, true>,%lld\n", Q, L, kRecords * Q * 16);
  REC(3, 1) REC(3, 2) REC(3, 4) REC(3, 16) REC(3, 64)
  REC(2, 1) REC(2, 2) REC(2, 4) REC(2, 16) REC(2, 64)
  CHECK(hipDeviceSynchronize());
  CHECK(hipFree(buf));
  return 0;
}

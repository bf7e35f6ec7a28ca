// Developer probe: what does a wave pay for gathering 128-byte records?
//   per-lane form   : every lane reads ITS record with eight 16-byte loads (64 cache lines per load instruction)
//   cooperative form: eight lanes share a record, one 16-byte piece each (8 cache lines per load instruction)
// hipcc --offload-arch=gfx950 -O3 tools/probe/gather_probe.hip -o tools/probe/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int PIECES>
__global__ void __launch_bounds__(256) k_lane(const uint4 *__restrict__ rec, const unsigned *__restrict__ idx, size_t n,
                                              unsigned *__restrict__ out) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 *r = rec + (size_t)idx[i] * 8;
    uint4 v[PIECES];
#pragma unroll
    for (int k = 0; k < PIECES; ++k) v[k] = r[k];
#pragma unroll
    for (int k = 0; k < PIECES; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// 8 lanes per record; the wave handles 8 records per load instruction, RPL load instructions in flight
template <int RPL>
__global__ void __launch_bounds__(256) k_coop(const uint4 *__restrict__ rec, const unsigned *__restrict__ idx, size_t n,
                                              unsigned *__restrict__ out) {
  unsigned acc = 0;
  const int piece = threadIdx.x & 7;
  const size_t grp = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3, ngrp = ((size_t)gridDim.x * blockDim.x) >> 3;
  for (size_t i = grp * RPL; i < n; i += ngrp * RPL) {
    uint4 v[RPL];
#pragma unroll
    for (int k = 0; k < RPL; ++k) v[k] = i + k < n ? rec[(size_t)idx[i + k] * 8 + piece] : uint4{0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < RPL; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char **argv) {
  const size_t n_gather = 1u << 22;  // records gathered per launch
  for (size_t table_mb : {4, 74, 1024}) {
    const size_t n_rec = table_mb * 1024 * 1024 / 128;
    uint4 *rec; unsigned *idx, *out;
    CK(hipMalloc(&rec, n_rec * 128)); CK(hipMalloc(&idx, n_gather * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(rec, 1, n_rec * 128));
    std::vector<unsigned> h(n_gather);
    std::mt19937 rng(1);
    for (auto &x : h) x = (unsigned)(rng() % n_rec);
    CK(hipMemcpy(idx, h.data(), n_gather * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char *name, auto launch) {
      launch(); CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
      printf("table %4zu MB  %-22s %8.1f us  %7.1f records/us  %6.2f records/us/CU  %6.0f GB/s\n", table_mb, name, ms * 1e3,
             n_gather / (ms * 1e3), n_gather / (ms * 1e3) / 256, n_gather * 128.0 / (ms * 1e6));
    };
    for (int occ : {2, 8}) {
      const dim3 g(256 * occ), b(256);
      printf("-- %d workgroups of 256 per CU\n", occ);
      time("per-lane 8 pieces", [&] { hipLaunchKernelGGL(k_lane<8>, g, b, 0, 0, rec, idx, n_gather, out); });
      time("per-lane 5 pieces", [&] { hipLaunchKernelGGL(k_lane<5>, g, b, 0, 0, rec, idx, n_gather, out); });
      time("per-lane 1 piece", [&] { hipLaunchKernelGGL(k_lane<1>, g, b, 0, 0, rec, idx, n_gather, out); });
      time("cooperative x1", [&] { hipLaunchKernelGGL(k_coop<1>, g, b, 0, 0, rec, idx, n_gather, out); });
      time("cooperative x4", [&] { hipLaunchKernelGGL(k_coop<4>, g, b, 0, 0, rec, idx, n_gather, out); });
    }
    CK(hipFree(rec)); CK(hipFree(idx)); CK(hipFree(out));
  }
  return 0;
}

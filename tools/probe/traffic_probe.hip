// tools/probe/traffic_probe.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the ACCESS PATTERNS of
// this backend (MI355X_MICROARCH.md states the x2 correction of FETCH_SIZE only for wide coalesced streaming reads and
// calls other widths uncalibrated).  Every kernel moves a KNOWN number of bytes over a buffer far larger than the 256 MB
// last-level cache:
//   k_stream_read      16 B per lane, coalesced, every byte once                       (the guide's calibrated case)
//   k_gather_rec128    one 128-byte record per lane through a random permutation, all 128 B read (8 x 16 B)
//                      = the candidate records (CRec) read by k_score3 / k_select / the tail through `perm`
//   k_gather_rec128_64 the same, first 64 B of each record only (the sweep: start, end, depths)
//   k_gather_sparse    128-byte records one every 2176 B (a staging list filled to 1/17) through a permutation
//   k_gather_seg128    128-byte records through a random index with REPEATS (a 6.4 MB table: the Seg gathers of
//                      k_tri_rows -- mostly cache hits; reported to show what the counter sees of them)
//   k_stream_write     16 B per lane, coalesced stores
//   k_scatter_rec128   full 128-byte records written to random positions (k_tri_rows / k_place style)
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/traffic_probe.hip -o tools/probe/traffic_probe
// run under rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE): tools/prof_traffic_calib.sh
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_stream_read(const uint4 *__restrict__ in, size_t n16, unsigned *__restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 v = in[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <int kUnits, size_t kStrideUnits, int kTag>  // kTag: distinct kernel names for the profiler
__global__ void k_gather(const uint4 *__restrict__ in, const unsigned *__restrict__ perm, size_t n, unsigned *__restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint4 *r = in + (size_t)perm[i] * kStrideUnits;
#pragma unroll
    for (int u = 0; u < kUnits; ++u) {
      const uint4 v = r[u];
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_stream_write(uint4 *__restrict__ out, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    out[i] = uint4{(unsigned)i, 1u, 2u, 3u};
}
__global__ void k_scatter_rec128(uint4 *__restrict__ out, const unsigned *__restrict__ perm, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint4 *r = out + (size_t)perm[i] * 8;
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = uint4{(unsigned)i, (unsigned)u, 2u, 3u};
  }
}

int main() {
  const size_t n_rec = 8u << 20;             // 8 Mi records x 128 B = 1 GiB
  const size_t bytes = n_rec * 128;
  uint4 *buf = nullptr, *sparse = nullptr;
  unsigned *perm = nullptr, *perm_sp = nullptr, *idx_seg = nullptr, *sink = nullptr;
  CHK(hipMalloc(&buf, bytes));
  CHK(hipMemset(buf, 1, bytes));
  const size_t n_sp = 1u << 20;              // 1 Mi records, one every 2176 B = 2.1 GiB
  CHK(hipMalloc(&sparse, n_sp * 2176));
  CHK(hipMemset(sparse, 1, n_sp * 2176));
  CHK(hipMalloc(&sink, 64));
  std::mt19937_64 rng(7);
  std::vector<unsigned> p(n_rec);
  std::iota(p.begin(), p.end(), 0u);
  std::shuffle(p.begin(), p.end(), rng);
  CHK(hipMalloc(&perm, 4 * n_rec));
  CHK(hipMemcpy(perm, p.data(), 4 * n_rec, hipMemcpyHostToDevice));
  std::vector<unsigned> ps(n_sp);
  std::iota(ps.begin(), ps.end(), 0u);
  std::shuffle(ps.begin(), ps.end(), rng);
  CHK(hipMalloc(&perm_sp, 4 * n_sp));
  CHK(hipMemcpy(perm_sp, ps.data(), 4 * n_sp, hipMemcpyHostToDevice));
  const size_t n_seg_tab = 50000, n_seg_reads = 4u << 20;  // 6.4 MB table, 4 Mi gathers
  std::vector<unsigned> is(n_seg_reads);
  for (auto &v : is) v = (unsigned)(rng() % n_seg_tab);
  CHK(hipMalloc(&idx_seg, 4 * n_seg_reads));
  CHK(hipMemcpy(idx_seg, is.data(), 4 * n_seg_reads, hipMemcpyHostToDevice));
  const dim3 grid(256 * 16), block(256);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_stream_read, grid, block, 0, 0, buf, bytes / 16, sink);
    hipLaunchKernelGGL((k_gather<8, 8, 0>), grid, block, 0, 0, buf, perm, n_rec, sink);        // k_gather_rec128
    hipLaunchKernelGGL((k_gather<4, 8, 1>), grid, block, 0, 0, buf, perm, n_rec, sink);        // k_gather_rec128_64
    hipLaunchKernelGGL((k_gather<8, 136, 2>), grid, block, 0, 0, sparse, perm_sp, n_sp, sink); // k_gather_sparse
    hipLaunchKernelGGL((k_gather<8, 8, 3>), grid, block, 0, 0, buf, idx_seg, n_seg_reads, sink); // k_gather_seg128
    hipLaunchKernelGGL(k_stream_write, grid, block, 0, 0, buf, bytes / 16);
    hipLaunchKernelGGL(k_scatter_rec128, grid, block, 0, 0, buf, perm, n_rec);
    CHK(hipDeviceSynchronize());
  }
  // known bytes per launch, in launch order (the profiling script pairs them with the counters)
  printf("known_bytes k_stream_read %zu\n", bytes);
  printf("known_bytes k_gather_rec128 %zu\n", n_rec * 128 + 4 * n_rec);
  printf("known_bytes k_gather_rec128_64 %zu\n", n_rec * 64 + 4 * n_rec);
  printf("known_bytes k_gather_sparse %zu\n", n_sp * 128 + 4 * n_sp);
  printf("known_bytes k_gather_seg128 %zu  (requested; the 6.4 MB table is cache resident)\n", n_seg_reads * 128 + 4 * n_seg_reads);
  printf("known_bytes k_stream_write %zu\n", bytes);
  printf("known_bytes k_scatter_rec128 %zu  (+ %zu index bytes read)\n", n_rec * 128, 4 * n_rec);
  return 0;
}

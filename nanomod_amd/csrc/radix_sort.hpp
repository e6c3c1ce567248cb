// Stable LSD radix sort of (uint64 key, uint32 value) pairs on the device — the sort behind nmod_rank_order / nmod_region_rank
// (Python's stable sorted() over the result records).  Hand-written (round 5; rounds 2-4 called rocPRIM's
// radix_sort_pairs here): eight passes of one byte, each
//   rs_hist_kernel     a block per tile of 2 048 pairs: the tile's digit histogram (LDS atomics) -> hist[digit][tile]
//   rs_scan_*          exclusive prefix sums over hist in (digit, tile) order: where each tile's run of each digit starts
//   rs_scatter_kernel  the tile again, eight rounds of 256 pairs in index order: a pair's place = start of its tile's run of its
//                      digit + the pairs of that digit in the tile's earlier rounds + those of lower lanes / waves in this round
//                      (eight ballots match the lanes of a wave that hold the same digit; a count per wave and digit in LDS)
// Equal digits keep their order (tile, round, thread = index order), so the pass is stable and so is the sort.  Eight passes:
// the result is back in the buffers it started in.  Not the timed hot path: 10 M pairs take a few milliseconds per pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nmod {

constexpr int kRsThreads = 256;
constexpr int kRsItems = 8;
constexpr int kRsTile = kRsThreads * kRsItems;                 // pairs per block
constexpr int kRsScanChunk = 4096;                             // histogram entries per block of the scan kernels

__global__ __launch_bounds__(kRsThreads) void rs_hist_kernel(const uint64_t* keys, int64_t n, int shift, int64_t ntiles, uint32_t* hist) {
  __shared__ unsigned h[256];
  h[threadIdx.x] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRsTile;
#pragma unroll
  for (int r = 0; r < kRsItems; ++r) {
    const int64_t i = base + r * kRsThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[(unsigned)(keys[i] >> shift) & 255u], 1u);
  }
  __syncthreads();
  hist[(int64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// ---- exclusive scan of m entries, in place: sums per chunk of 4 096, a scan of those by one block, then each chunk
__device__ __forceinline__ unsigned rs_block_exscan_256(unsigned v, unsigned* sh, unsigned& total) {   // 256 threads; sh[256]
  sh[threadIdx.x] = v;
  __syncthreads();
#pragma unroll
  for (int d = 1; d < 256; d <<= 1) {
    const unsigned add = threadIdx.x >= (unsigned)d ? sh[threadIdx.x - d] : 0u;
    __syncthreads();
    sh[threadIdx.x] += add;
    __syncthreads();
  }
  total = sh[255];
  const unsigned incl = sh[threadIdx.x];
  __syncthreads();
  return incl - v;
}
__global__ __launch_bounds__(256) void rs_scan_reduce_kernel(const uint32_t* g, int64_t m, uint32_t* bsum) {
  __shared__ unsigned sh[256];
  const int64_t base = (int64_t)blockIdx.x * kRsScanChunk + (int64_t)threadIdx.x * 16;
  unsigned s = 0u;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += base + e < m ? g[base + e] : 0u;
  unsigned total;
  rs_block_exscan_256(s, sh, total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void rs_scan_tops_kernel(uint32_t* bsum, int64_t nb) {          // one block
  __shared__ unsigned sh[256];
  unsigned carry = 0u;
  for (int64_t b0 = 0; b0 < nb; b0 += 256) {
    const int64_t i = b0 + threadIdx.x;
    const unsigned v = i < nb ? bsum[i] : 0u;
    unsigned total;
    const unsigned ex = rs_block_exscan_256(v, sh, total);
    if (i < nb) bsum[i] = carry + ex;
    carry += total;
  }
}
__global__ __launch_bounds__(256) void rs_scan_apply_kernel(uint32_t* g, int64_t m, const uint32_t* bsum) {
  __shared__ unsigned sh[256];
  const int64_t base = (int64_t)blockIdx.x * kRsScanChunk + (int64_t)threadIdx.x * 16;
  unsigned v[16], s = 0u;
#pragma unroll
  for (int e = 0; e < 16; ++e) { v[e] = base + e < m ? g[base + e] : 0u; s += v[e]; }
  unsigned total;
  unsigned run = bsum[blockIdx.x] + rs_block_exscan_256(s, sh, total);
#pragma unroll
  for (int e = 0; e < 16; ++e) { if (base + e < m) g[base + e] = run; run += v[e]; }
}

__global__ __launch_bounds__(kRsThreads) void rs_scatter_kernel(const uint64_t* keys, const uint32_t* vals, int64_t n, int shift, int64_t ntiles,
                                                                 const uint32_t* offs, uint64_t* keys_out, uint32_t* vals_out) {
  __shared__ unsigned start[256];                              // where the next pair of each digit goes
  __shared__ unsigned wcnt[kRsThreads / 64][256];              // this round: pairs of each digit per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  start[threadIdx.x] = offs[(int64_t)threadIdx.x * ntiles + blockIdx.x];
#pragma unroll
  for (int w = 0; w < kRsThreads / 64; ++w) wcnt[w][threadIdx.x] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kRsTile;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll 1
  for (int r = 0; r < kRsItems; ++r) {
    const int64_t i = base + r * kRsThreads + threadIdx.x;
    const bool valid = i < n;
    const uint64_t key = valid ? keys[i] : 0ull;
    const uint32_t val = valid ? vals[i] : 0u;
    const unsigned d = (unsigned)(key >> shift) & 255u;
    unsigned long long peers = __ballot(valid);               // the lanes of this wave that hold the same digit
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = ((d >> b) & 1u) != 0u;
      const unsigned long long bm = __ballot(bit);
      peers &= bit ? bm : ~bm;
    }
    const unsigned before_lane = (unsigned)__popcll(peers & lt);
    if (valid && before_lane == 0u) wcnt[wave][d] = (unsigned)__popcll(peers);
    __syncthreads();
    if (valid) {
      unsigned at = start[d] + before_lane;
      for (int w = 0; w < wave; ++w) at += wcnt[w][d];
      keys_out[at] = key; vals_out[at] = val;
    }
    __syncthreads();
    {
      unsigned tot = 0u;
#pragma unroll
      for (int w = 0; w < kRsThreads / 64; ++w) { tot += wcnt[w][threadIdx.x]; wcnt[w][threadIdx.x] = 0u; }
      start[threadIdx.x] += tot;
    }
    __syncthreads();
  }
}

// Scratch the sort needs beside the two pairs of buffers: the histogram (256 entries per tile) and the scan's chunk sums.
inline int64_t rs_tiles(int64_t n) { return (n + kRsTile - 1) / kRsTile; }
inline size_t rs_scratch_bytes(int64_t n) {
  const int64_t m = 256 * rs_tiles(n);
  return (size_t)(m + (m + kRsScanChunk - 1) / kRsScanChunk + 1) * 4;
}
// keys / vals: the pairs, sorted in place (ascending keys, stable); keys_tmp / vals_tmp: as large.  n < 2^31.
inline hipError_t rs_sort_pairs(uint64_t* keys, uint32_t* vals, uint64_t* keys_tmp, uint32_t* vals_tmp, int64_t n, void* scratch, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  const int64_t ntiles = rs_tiles(n), m = 256 * ntiles, nb = (m + kRsScanChunk - 1) / kRsScanChunk;
  uint32_t* hist = static_cast<uint32_t*>(scratch);
  uint32_t* bsum = hist + m;
  for (int pass = 0; pass < 8; ++pass) {
    const uint64_t* ksrc = (pass & 1) ? keys_tmp : keys; const uint32_t* vsrc = (pass & 1) ? vals_tmp : vals;
    uint64_t* kdst = (pass & 1) ? keys : keys_tmp; uint32_t* vdst = (pass & 1) ? vals : vals_tmp;
    hipLaunchKernelGGL(rs_hist_kernel, dim3((unsigned)ntiles), dim3(kRsThreads), 0, stream, ksrc, n, 8 * pass, ntiles, hist);
    hipLaunchKernelGGL(rs_scan_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, (const uint32_t*)hist, m, bsum);
    hipLaunchKernelGGL(rs_scan_tops_kernel, dim3(1), dim3(256), 0, stream, bsum, nb);
    hipLaunchKernelGGL(rs_scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, stream, hist, m, (const uint32_t*)bsum);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3((unsigned)ntiles), dim3(kRsThreads), 0, stream, ksrc, vsrc, n, 8 * pass, ntiles,
                       (const uint32_t*)hist, kdst, vdst);
  }
  return hipGetLastError();
}

}  // namespace nmod

// Host-side launch interface of K1 (one translation unit per dtype x tests pair).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nmod {

struct RankStatsArgs;

constexpr int kNumSizeClasses = 6;                 // R = 1, 2, 4, 8, 16, 32 registers per lane per group
inline int size_class_of(int64_t n) {              // smallest class with 64 * R >= n; -1 if too large
  for (int c = 0; c < kNumSizeClasses; ++c) if (n <= (64LL << c)) return c;
  return -1;
}
inline size_t rank_stats_lds_bytes(int c0, int c1, bool all) {
  size_t keys = (64u << c0) + 4 + (64u << c1) + 4;  // words per wave (kLdsPad = 4)
  return (all ? 2 : 1) * keys * 4 /*bytes*/ * 4 /*waves per block*/;
}

hipError_t launch_rank_stats_d0_a0(int c0, int c1, int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a);
hipError_t launch_rank_stats_d0_a1(int c0, int c1, int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a);
hipError_t launch_rank_stats_d1_a0(int c0, int c1, int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a);
hipError_t launch_rank_stats_d1_a1(int c0, int c1, int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a);

}  // namespace nmod

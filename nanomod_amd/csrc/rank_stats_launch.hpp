// Host-side launch interface of K1 (one translation unit per dtype x tests pair).
//
// Size classes.  A group of n samples needs capacity 64 << c (c = 0..5: 64 .. 2048).
//   general class  c0 * 6 + c1      (0..35): min(c0, c1) <= 2 -> rank_hist_kernel<1 << min, 64, dtype, WIDE> (the smaller
//                  group sorted over the 64 lanes, the larger one streamed; rank_hist.hpp); otherwise
//                  rank_pair_kernel<1<<c0, 1<<c1> (rank_all.hpp), both groups sorted, 64 lanes per group;
//   packed class   36 + cm          (36..40): rank_hist_kernel (rank_hist.hpp), both groups in capacity 64 << cm,
//                  used when max(c0,c1) = cm <= 4 and min(c0,c1) >= cm - 1:
//                  cm 0..1 -> (R, LG) = (8,8) (16,8): eight positions per wave; cm 2 -> (16,16): four; cm 3 -> (16,32): two;
//                  cm 4 -> (16,64): one (16 keys per lane keep every form at four waves per SIMD).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nmod {

struct RankStatsArgs;

constexpr int kNumSizeClasses = 6;                 // capacities 64, 128, ..., 2048
constexpr int kNumGeneralClasses = kNumSizeClasses * kNumSizeClasses;
constexpr int kNumPackedClasses = 5;
constexpr int kNumKsClasses = 6;                   // KS-only form (ks_rank.hpp): class of the SMALLER group
constexpr int kKsClassBase = kNumGeneralClasses + kNumPackedClasses;    // 41
constexpr int kNumClasses = kKsClassBase + kNumKsClasses;               // 47 (+ the large-position classes) <= kClassStride (56)
// large positions (at least one group beyond 2 048 samples; binned by classify_kernel, nanomod_hip.hip)
constexpr int kBigClass = kNumClasses;             // 47: big_rank_kernel (big_rank.hpp)
constexpr int kBigHistClass = kNumClasses + 1;     // 48: all tests, big_hist_kernel (smaller group <= 1 024, larger <= 4 096)
constexpr int kWideBigBase = kNumClasses + 2;      // 49 + cs: all tests, smaller group in class cs <= 2 (<= 256), larger 2 049 .. 4 096:
constexpr int kNumWideBig = 3;                     //          rank_hist_kernel WIDE with two hash passes (rank_hist.hpp)
constexpr int kWideBigMaxQ = 4096;
constexpr int kNumPairs = kWideBigBase + kNumWideBig;                   // 52 class lists

__host__ __device__ inline int size_class_of(int64_t n) {   // smallest class with 64 << c >= n; 6 if too large
  int c = 0;
  while (c < kNumSizeClasses && n > (64LL << c)) ++c;
  return c;
}
__host__ __device__ inline int launch_class_of(int c0, int c1) {
  int cm = c0 > c1 ? c0 : c1, cl = c0 > c1 ? c1 : c0;
  if (cm <= 4 && cl >= cm - 1) return kNumGeneralClasses + cm;
  return c0 * kNumSizeClasses + c1;
}
// general classes served by the WIDE form of rank_hist_kernel: the smaller group fits 256 sorted samples
inline bool wide_class(int cls) {
  if (cls >= kWideBigBase) return cls < kWideBigBase + kNumWideBig;
  return cls < kNumGeneralClasses && (cls / kNumSizeClasses <= 2 || cls % kNumSizeClasses <= 2);
}
// 32-bit words of a wave's tie table in the WIDE form (rank_hist.hpp): see wide_table_words.
#ifndef NMOD_WIDE_I16_WORDS
#define NMOD_WIDE_I16_WORDS 2048                   // 8 192 values, four blocks per CU (1 912 words = five blocks: measured 10 % slower)
#endif
__host__ __device__ constexpr int wide_table_words(int cls, int dtype) {
  // int16: NMOD_WIDE_I16_WORDS (counters over 8 192 values).  float32 (round 4): 2 048 words for every class — the two bitmaps of
  // the bitmap form (B1's words double as the exact table of the few samples on shared bits), the counters of the grid form
  // (8 192 values); the multiset hash that sized the table by the class of the larger group (up to 4 100 words: two blocks per
  // CU for groups beyond 2 048 samples) is gone — a position with more shared bits than the exact table takes is finished by
  // wide_redo_kernel.  Every class of R = 1, 2 fits four blocks per CU (+6.5 % on configs[4] over 3 068 / 4 100 words).
  (void)cls;
  return dtype == 1 ? NMOD_WIDE_I16_WORDS : 2048;
}
__host__ __device__ constexpr int wide_table_slots(int words) {
  switch (words) {
    case 1024: return 1021;
    case 1600: return 1597;
    case 3068: return 3067;
    case 2056: return 2053;
    case 4100: return 4099;
    case 2048: return 2039;
    case 4096: return 4093;
    case 512: return 509;
    case 256: return 251;
    case 128: return 127;
    default: return words > 8 ? (words - 1) | 1 : 7;     // (not reached: every size above has its prime)
  }
}
inline int wide_class_of_s(int cls) {              // capacity class of the smaller group of a WIDE class
  if (cls >= kWideBigBase) return cls - kWideBigBase;
  return cls / kNumSizeClasses < cls % kNumSizeClasses ? cls / kNumSizeClasses : cls % kNumSizeClasses;
}
// capacity 64 << cs sorted in (R, LG) = (8,8) (16,8) (16,16) (32,16) (32,32) (32,64): the packed all-tests classes 0..4;
// the KS-only classes take (16,32) for cs = 3 (below)
inline int ks_lanes_per_group(int cs) { return cs <= 1 ? 8 : (cs == 2 ? 16 : (8 << (cs - 2))); }
inline int ks_positions_per_wave(int cs) { return 64 / ks_lanes_per_group(cs); }
// KS-only form: capacity 512 (cs = 3) as 16 keys x 32 lanes — two positions per wave at four waves per SIMD (117 registers,
// 8.4 KB of LDS per wave) instead of 32 x 16 with four positions at two waves per SIMD (176 registers, 17.4 KB); capacity
// 1 024 (cs = 4) as 16 x 64 for the same reason
inline int ksonly_lanes_per_group(int cs) { return cs == 3 ? 32 : (cs == 4 ? 64 : ks_lanes_per_group(cs)); }
inline int ksonly_positions_per_wave(int cs) { return 64 / ksonly_lanes_per_group(cs); }
// packed all-tests classes (rank_hist.hpp): the same (R, LG) per capacity as the KS-only classes 0..4
inline int packed_lanes_per_group(int cm) { return cm == 3 ? 32 : (cm == 4 ? 64 : ks_lanes_per_group(cm)); }   // (capacities 512 / 1 024 as 16 x 32 / 16 x 64, like the KS-only forms)
inline int packed_positions_per_wave(int cm) { return 64 / packed_lanes_per_group(cm); }
static inline size_t rank_stats_lds_bytes(int cls, bool all, int dtype) {
  size_t words;
  if (wide_class(cls)) {
    // rank_hist_kernel WIDE (rank_hist.hpp): keys + bins of S (rounded to 16 bytes) + the wave's tie table, + two doubles
    const size_t R = (size_t)1 << wide_class_of_s(cls);
    const size_t w = ((2 * R * 65 + 3) & ~(size_t)3) + (size_t)wide_table_words(cls, dtype) + (dtype == 1 ? 0 : 128);   // float32: + kWideList (rank_hist.hpp)
    return w * 4 * 4 + 16;
  } else if (cls >= kKsClassBase) {
    int cs = cls - kKsClassBase;
    size_t LG = (size_t)ksonly_lanes_per_group(cs), R = (64u << cs) / LG;
    size_t w = 2 * R * (LG + 1);                                             // ks_rank_pos_words (ks_rank.hpp)
    if (LG <= 16) while ((w & 31) != LG) ++w;
    return (size_t)ksonly_positions_per_wave(cs) * w * 4 * 4 + 16;              // bytes, 4 waves per block, + two doubles (ks_rank.hpp: recip)
  } else if (cls >= kNumGeneralClasses) {
    int cm = cls - kNumGeneralClasses;
    size_t LG = (size_t)packed_lanes_per_group(cm), R = (64u << cm) / LG;
    size_t w = 2 * R * (LG + 1);                                             // ks_rank_pos_words (ks_rank.hpp)
    if (LG <= 16) while ((w & 31) != LG) ++w;
    return (size_t)packed_positions_per_wave(cm) * w * 4 * 4 + 16;             // + two doubles (rank_hist.hpp: recip)
  } else {
    // rank_pair_kernel (rank_all.hpp): keys + runs of both groups, R x 65 words each
    words = 2 * 65 * ((1u << (cls / kNumSizeClasses)) + (1u << (cls % kNumSizeClasses)));
  }
  (void)all;
  return words * 4 /*bytes*/ * 4 /*waves per block*/;
}

// The counting form for any coverage (rank_count_wide.hpp): which launch classes can take it, and with how many registers per lane
// for the smaller group (index into {1, 2, 4, 8, 16}: 64 ... 1 024 samples; 5: neither group in registers, rank_count_value.hpp); -1: not this class (the 256-capacity packed class has
// its own counting form, rank_count.hpp; the small packed classes run eight positions per wave already)
__host__ __device__ inline int count_wide_rs_index(int cls) {
  if (cls >= kWideBigBase && cls < kWideBigBase + kNumWideBig) return cls - kWideBigBase;           // smaller group in class 0..2
  if (cls >= kNumGeneralClasses && cls < kKsClassBase) { const int cm = cls - kNumGeneralClasses; return cm == 3 ? 3 : (cm == 4 ? 4 : -1); }
  if (cls >= 0 && cls < kNumGeneralClasses) { const int c0 = cls / kNumSizeClasses, c1 = cls % kNumSizeClasses, s = c0 < c1 ? c0 : c1; return s; }   // (5: both groups of 1 025 .. 2 048 samples — the value-domain form, rank_count_value.hpp)
  if (cls >= kKsClassBase && cls < kKsClassBase + kNumKsClasses) { const int cs = cls - kKsClassBase; return cs <= 5 ? cs : -1; }   // KS-only: class of the smaller group (5: both above 1 024, the value-domain form)
  return -1;
}
// KS-only mode: a class holds every position whose SMALLER group has the class's capacity; the form takes those whose larger group has at
// least this many samples (200 v 200 and below stay with ks_rank_kernel, four or eight positions per wave)
constexpr int kCwKsMinQ = 320;
struct CountWideWs { int32_t* gates; int32_t* work_list; int32_t* work_meta; };
// once per batch: one probe block per class in `classes` (gates[class]), then the work lists as copies of the class lists
hipError_t launch_count_wide_prepare_d0_a1(const int* classes, int nclasses, hipStream_t s, const RankStatsArgs& a, const CountWideWs& w);
hipError_t launch_count_wide_prepare_d1_a1(const int* classes, int nclasses, hipStream_t s, const RankStatsArgs& a, const CountWideWs& w);
hipError_t launch_count_wide_prepare_d0_a0(const int* classes, int nclasses, hipStream_t s, const RankStatsArgs& a, const CountWideWs& w);   // (KS-only)
hipError_t launch_count_wide_prepare_d1_a0(const int* classes, int nclasses, hipStream_t s, const RankStatsArgs& a, const CountWideWs& w);
// then, before the classes' sorting launches: rank_count_wide_kernel over every class whose gate is set; it appends what it hands on
// to the work lists itself (value_class: a class of index 5 is among them, rank_count_value_kernel follows in a launch of its own)
hipError_t launch_count_wide_run_d0_a1(int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a, const CountWideWs& w, bool value_class);
hipError_t launch_count_wide_run_d1_a1(int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a, const CountWideWs& w, bool value_class);
hipError_t launch_count_wide_run_d0_a0(int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a, const CountWideWs& w, bool value_class);
hipError_t launch_count_wide_run_d1_a0(int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a, const CountWideWs& w, bool value_class);

hipError_t launch_rank_stats_d0_a0(int cls, int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a);
hipError_t launch_rank_stats_d0_a1(int cls, int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a);
hipError_t launch_rank_stats_d1_a0(int cls, int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a);
hipError_t launch_rank_stats_d1_a1(int cls, int num_cus, int64_t work_items, hipStream_t s, const RankStatsArgs& a);
// the experiment macros each of the four translation units was compiled with (build_info.hpp)
const char* rank_stats_build_flags_d0_a0();
const char* rank_stats_build_flags_d0_a1();
const char* rank_stats_build_flags_d1_a0();
const char* rank_stats_build_flags_d1_a1();

}  // namespace nmod

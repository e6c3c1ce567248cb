// Instantiates K1 for every (R0, R1) size class of one (dtype, tests) pair.
// Built four times by the Makefile: -DNMOD_INST_DTYPE={0,1} -DNMOD_INST_ALL={0,1}.
#include <algorithm>
#include <atomic>
#include "rank_stats.hpp"
#include "rank_stats_packed.hpp"
#include "ks_rank.hpp"
#include "rank_all.hpp"
#include "rank_hist.hpp"
#include "rank_count.hpp"
#include "rank_count_wide.hpp"
#include "rank_count_value.hpp"
#include "rank_stats_launch.hpp"
#include "build_info.hpp"

#ifndef NMOD_INST_DTYPE
#error "define NMOD_INST_DTYPE and NMOD_INST_ALL"
#endif

namespace nmod {

namespace {
constexpr int DT = NMOD_INST_DTYPE;
constexpr bool ALL = NMOD_INST_ALL != 0;
using KernelFn = void (*)(RankStatsArgs);

#if NMOD_INST_ALL
// all-tests builds: rank_hist_kernel for same-class positions, rank_pair_kernel (one position per wave) for the rest
template <int C0, int C1>
constexpr KernelFn kernel_of() { return rank_pair_kernel<(1 << C0), (1 << C1), DT>; }

template <int C0>
KernelFn pick1(int c1) {
  switch (c1) {
    case 0: return kernel_of<C0, 0>();
    case 1: return kernel_of<C0, 1>();
    case 2: return kernel_of<C0, 2>();
    case 3: return kernel_of<C0, 3>();
    case 4: return kernel_of<C0, 4>();
    default: return kernel_of<C0, 5>();
  }
}
KernelFn pick(int c0, int c1) {
  switch (c0) {
    case 0: return pick1<0>(c1);
    case 1: return pick1<1>(c1);
    case 2: return pick1<2>(c1);
    case 3: return pick1<3>(c1);
    case 4: return pick1<4>(c1);
    default: return pick1<5>(c1);
  }
}
KernelFn pick_wide(int cmin) {
  switch (cmin) {
    case 0: return rank_hist_kernel<1, 64, DT, true>;
    case 1: return rank_hist_kernel<2, 64, DT, true>;
    default: return rank_hist_kernel<4, 64, DT, true>;
  }
}
KernelFn pick_packed(int cm, bool after_count) {
  if (after_count) return rank_hist_kernel<16, 16, DT, false, true>;     // (cm == 2: behind rank_count_kernel)
  switch (cm) {
    case 0: return rank_hist_kernel<8, 8, DT>;
    case 1: return rank_hist_kernel<16, 8, DT>;
    case 2: return rank_hist_kernel<16, 16, DT>;
    case 3: return rank_hist_kernel<16, 32, DT>;
    default: return rank_hist_kernel<16, 64, DT>;
  }
}
#else
// KS-only builds carry only the ks_rank kernels
KernelFn pick_ks(int cs, bool flags) {
#if NMOD_INST_DTYPE == 0
  if (flags) {                 // float32 images of float64 samples: report ties (ks_rank.hpp)
    switch (cs) {
      case 0: return ks_rank_kernel<8, 8, DT, true>;
      case 1: return ks_rank_kernel<16, 8, DT, true>;
      case 2: return ks_rank_kernel<16, 16, DT, true>;
      case 3: return ks_rank_kernel<16, 32, DT, true>;
      case 4: return ks_rank_kernel<16, 64, DT, true>;
      default: return ks_rank_kernel<32, 64, DT, true>;
    }
  }
#endif
  (void)flags;
  switch (cs) {
    case 0: return ks_rank_kernel<8, 8, DT>;
    case 1: return ks_rank_kernel<16, 8, DT>;
    case 2: return ks_rank_kernel<16, 16, DT>;
    case 3: return ks_rank_kernel<16, 32, DT>;
    case 4: return ks_rank_kernel<16, 64, DT>;
    default: return ks_rank_kernel<32, 64, DT>;
  }
}
#endif
}  // namespace

#define NMOD_CAT2(a, b) a##b
#define NMOD_CAT(a, b) NMOD_CAT2(a, b)
#define NMOD_LAUNCH_NAME NMOD_CAT(NMOD_CAT(launch_rank_stats_d, NMOD_INST_DTYPE), NMOD_CAT(_a, NMOD_INST_ALL))
#define NMOD_FLAGS_NAME NMOD_CAT(NMOD_CAT(rank_stats_build_flags_d, NMOD_INST_DTYPE), NMOD_CAT(_a, NMOD_INST_ALL))

// the experiment macros THIS translation unit's kernels were compiled with (nmod_build_info)
const char* NMOD_FLAGS_NAME() { return NMOD_BUILD_FLAGS; }

hipError_t NMOD_LAUNCH_NAME(int cls, int num_cus, int64_t work_items, hipStream_t stream,
                            const RankStatsArgs& args) {
  const bool ks = cls >= kKsClassBase && cls < kNumClasses;
  const bool packed = cls >= kNumGeneralClasses && cls < kKsClassBase;
  KernelFn fn = nullptr;
#if NMOD_INST_ALL
  if (ks) return hipErrorInvalidValue;
  const bool wide = wide_class(cls);
  // capacity-256 positions with all tests: the counting form first (rank_count.hpp) when the caller provided its gate and flags
  const bool counting = packed && cls - kNumGeneralClasses == 2 && args.cnt_gate != nullptr && args.cnt_done != nullptr;
  fn = packed ? pick_packed(cls - kNumGeneralClasses, counting)
       : wide ? pick_wide(wide_class_of_s(cls))
              : pick(cls / kNumSizeClasses, cls % kNumSizeClasses);
#else
  if (!ks) return hipErrorInvalidValue;
  fn = pick_ks(cls - kKsClassBase, args.tied != nullptr);
#endif
  const size_t lds = rank_stats_lds_bytes(cls, ALL, DT);
  if (ks || packed) {
    const int pw = ks ? ksonly_positions_per_wave(cls - kKsClassBase) : packed_positions_per_wave(cls - kNumGeneralClasses);
    work_items = (work_items + pw - 1) / pw;
  }
  // the dynamic-LDS attribute and the occupancy of a kernel are looked up once per (device, class), not on every launch
  // (atomics: concurrent first launches of a class both run the queries and store the same number)
  static std::atomic<int> per_cu_cache[64][2 * kClassStride];     // [class] and [kClassStride + class] for the FLAGS instances
  int dev = 0;
  const bool cacheable = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
#if NMOD_INST_ALL
  const int slot_id = cls + (counting ? kClassStride : 0);
#else
  const int slot_id = cls + ((ks && args.tied) ? kClassStride : 0);
#endif
  int per_cu = cacheable ? per_cu_cache[dev][slot_id].load(std::memory_order_relaxed) : 0;
  if (per_cu <= 0) {
    // (one WIDE instance serves every class of the larger group: its limit is that of the largest)
    size_t lds_limit = lds;
#if NMOD_INST_ALL
    if (wide) lds_limit = rank_stats_lds_bytes(kWideBigBase + wide_class_of_s(cls), ALL, DT);
#endif
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_limit);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * kWavesPerBlock, lds);
    if (e != hipSuccess) return e;
    if (per_cu < 1) return hipErrorLaunchOutOfResources;     // (never launch a block that cannot get its LDS: its table walks would not end)
    if (cacheable) per_cu_cache[dev][slot_id].store(per_cu, std::memory_order_relaxed);
  }
#if NMOD_INST_ALL
  if (counting) {
    // probe (one block: is the batch event-like?) -> rank_count_kernel (exits at once when it is not) -> the AFTER instance below
    CntProbeArgs pa;
    pa.sig0 = args.sig0; pa.sig1 = args.sig1; pa.off0 = args.off0; pa.off1 = args.off1; pa.stride0 = args.stride0; pa.stride1 = args.stride1;
    pa.npos = args.npos; pa.pos_list = args.pos_list; pa.class_meta = args.class_meta; pa.class_id = args.class_id; pa.dtype = DT; pa.gate = args.cnt_gate;
    // (float32 keys of float64 samples — args.tied is set — are whole numbers where the samples sit on the 0.001 grid: DTYPE 2)
#if NMOD_INST_DTYPE == 0
    const bool int_keys = args.tied != nullptr;
    if (int_keys) hipLaunchKernelGGL(cnt_probe_kernel<2>, dim3(1), dim3(1024), 0, stream, pa);
    else hipLaunchKernelGGL(cnt_probe_kernel<0>, dim3(1), dim3(1024), 0, stream, pa);
    static std::atomic<int> cnt_per_cu[64][2];
    KernelFn cfn = int_keys ? (KernelFn)rank_count_kernel<2> : (KernelFn)rank_count_kernel<0>;
    std::atomic<int>& cpc_slot = cnt_per_cu[cacheable ? dev : 0][int_keys ? 1 : 0];
#else
    hipLaunchKernelGGL(cnt_probe_kernel<DT>, dim3(1), dim3(1024), 0, stream, pa);
    static std::atomic<int> cnt_per_cu[64][1];
    KernelFn cfn = rank_count_kernel<DT>;
    std::atomic<int>& cpc_slot = cnt_per_cu[cacheable ? dev : 0][0];
#endif
    const size_t clds = rank_count_lds_bytes();
    int cpc = cacheable ? cpc_slot.load(std::memory_order_relaxed) : 0;
    if (cpc <= 0) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(cfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds);
      if (e != hipSuccess) return e;
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&cpc, cfn, 64 * kWavesPerBlock, clds);
      if (e != hipSuccess) return e;
      if (cpc < 1) return hipErrorLaunchOutOfResources;
      if (cacheable) cpc_slot.store(cpc, std::memory_order_relaxed);
    }
    int64_t cblocks = std::min<int64_t>((work_items + kWavesPerBlock - 1) / kWavesPerBlock, (int64_t)num_cus * cpc);
    if (cblocks < 1) cblocks = 1;
    hipLaunchKernelGGL(cfn, dim3((unsigned)cblocks), dim3(64 * kWavesPerBlock), clds, stream, args);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
#endif
  int64_t blocks = (work_items + kWavesPerBlock - 1) / kWavesPerBlock;
  int64_t cap = (int64_t)num_cus * per_cu;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
#if NMOD_INST_ALL
  if (counting) {
    // gate clear: the plain instance over the whole list; gate set: the AFTER instance over what the counting form left
    RankStatsArgs a2 = args;
    a2.cnt_mode = 1;
    KernelFn plain = pick_packed(cls - kNumGeneralClasses, false);
    static std::atomic<int> plain_ready[64];
    if (!cacheable || plain_ready[dev].load(std::memory_order_relaxed) == 0) {
      const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(plain), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      if (cacheable) plain_ready[dev].store(1, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(plain, dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds, stream, a2);
    a2.cnt_mode = 2;
    hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds, stream, a2);
    return hipGetLastError();
  }
#endif
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds, stream, args);
  return hipGetLastError();
}

}  // namespace nmod

#define NMOD_CW_PREP_NAME NMOD_CAT(NMOD_CAT(launch_count_wide_prepare_d, NMOD_INST_DTYPE), NMOD_CAT(_a, NMOD_INST_ALL))
#define NMOD_CW_RUN_NAME NMOD_CAT(NMOD_CAT(launch_count_wide_run_d, NMOD_INST_DTYPE), NMOD_CAT(_a, NMOD_INST_ALL))
namespace nmod {

hipError_t NMOD_CW_PREP_NAME(const int* classes, int nclasses, hipStream_t stream, const RankStatsArgs& a, const CountWideWs& w) {
  if (nclasses <= 0) return hipSuccess;
  if (nclasses > kClassStride) return hipErrorInvalidValue;        // (cls[] / max_s[] / the segs area hold one entry per class)
  CntWideProbeArgs pa;
  pa.sig0 = a.sig0; pa.sig1 = a.sig1; pa.off0 = a.off0; pa.off1 = a.off1; pa.stride0 = a.stride0; pa.stride1 = a.stride1; pa.npos = a.npos;
  pa.pos_list = a.pos_list; pa.class_meta = a.class_meta; pa.nclasses = nclasses; pa.gate = w.gates; pa.segs = w.gates + kClassStride; pa.work_meta = w.work_meta; pa.min_q = NMOD_INST_ALL ? 0 : kCwKsMinQ;
  for (int i = 0; i < nclasses && i < kClassStride; ++i) { pa.cls[i] = classes[i]; pa.max_s[i] = 64 << count_wide_rs_index(classes[i]); }
  hipError_t e = hipMemsetAsync(w.gates, 0, kClassStride * 4, stream);
  if (e != hipSuccess) return e;
#if NMOD_INST_DTYPE == 0
  if (a.tied != nullptr) hipLaunchKernelGGL(HIP_KERNEL_NAME(cnt_wide_probe_kernel<2, NMOD_INST_ALL == 0>), dim3((unsigned)nclasses), dim3(1024), 0, stream, pa);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(cnt_wide_probe_kernel<0, NMOD_INST_ALL == 0>), dim3((unsigned)nclasses), dim3(1024), 0, stream, pa);
#else
  hipLaunchKernelGGL(HIP_KERNEL_NAME(cnt_wide_probe_kernel<1, NMOD_INST_ALL == 0>), dim3((unsigned)nclasses), dim3(1024), 0, stream, pa);
#endif
  return hipGetLastError();
}

hipError_t NMOD_CW_RUN_NAME(int num_cus, int64_t work_items, hipStream_t stream, const RankStatsArgs& a, const CountWideWs& w, bool value_class) {
  typedef void (*CwFn)(CntWideArgs);
#if NMOD_INST_DTYPE == 0
  const bool int_keys = a.tied != nullptr;
  CwFn fn = int_keys ? (CwFn)rank_count_wide_kernel<2, NMOD_INST_ALL == 0> : (CwFn)rank_count_wide_kernel<0, NMOD_INST_ALL == 0>;
  const int slot = int_keys ? 1 : 0;
#else
  CwFn fn = (CwFn)rank_count_wide_kernel<1, NMOD_INST_ALL == 0>;
  const int slot = 0;
#endif
  static std::atomic<int> per_cu[64][2];
  int dev = 0;
  const bool cacheable = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
  const size_t lds = rank_count_wide_lds_bytes();
  int pc = cacheable ? per_cu[dev][slot].load(std::memory_order_relaxed) : 0;
  if (pc <= 0) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, fn, 64 * kWavesPerBlock, lds);
    if (e != hipSuccess) return e;
    if (pc < 1) return hipErrorLaunchOutOfResources;
    if (cacheable) per_cu[dev][slot].store(pc, std::memory_order_relaxed);
  }
  CntWideArgs ca;
  ca.rs = a; ca.gates = w.gates; ca.segs = w.gates + kClassStride; ca.work_list = w.work_list; ca.work_meta = w.work_meta;
  int64_t blocks = std::min<int64_t>((work_items + kWavesPerBlock - 1) / kWavesPerBlock, (int64_t)num_cus * pc);
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(64 * kWavesPerBlock), lds, stream, ca);
  if (value_class) {
    // the class whose groups both hold more than 1 024 samples: the value-domain form (rank_count_value.hpp), a launch of its own
#if NMOD_INST_DTYPE == 0
    CwFn vfn = int_keys ? (CwFn)rank_count_value_kernel<2, NMOD_INST_ALL == 0> : (CwFn)rank_count_value_kernel<0, NMOD_INST_ALL == 0>;
#else
    CwFn vfn = (CwFn)rank_count_value_kernel<1, NMOD_INST_ALL == 0>;
#endif
    static std::atomic<int> vper_cu[64][2];
    const size_t vlds = rank_count_value_lds_bytes();
    int vpc = cacheable ? vper_cu[dev][slot].load(std::memory_order_relaxed) : 0;
    if (vpc <= 0) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)vlds);
      if (e != hipSuccess) return e;
      e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&vpc, vfn, 64 * kWavesPerBlock, vlds);
      if (e != hipSuccess) return e;
      if (vpc < 1) return hipErrorLaunchOutOfResources;
      if (cacheable) vper_cu[dev][slot].store(vpc, std::memory_order_relaxed);
    }
    int64_t vblocks = std::min<int64_t>((work_items + kWavesPerBlock - 1) / kWavesPerBlock, (int64_t)num_cus * vpc);
    if (vblocks < 1) vblocks = 1;
    hipLaunchKernelGGL(vfn, dim3((unsigned)vblocks), dim3(64 * kWavesPerBlock), vlds, stream, ca);
  }
  return hipGetLastError();
}

}  // namespace nmod


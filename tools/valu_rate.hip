// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the instruction kinds K1's
// register sort is made of, at 8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../nanomod_amd/csrc/wave_ops.hpp"
#include "../nanomod_amd/csrc/packed_sort_i16.hpp"
using namespace nmod;

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  int lane = threadIdx.x & 63;
  float x[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = seed * (float)(lane * 17 + r * 3 + 1);
  float c = (lane & 1) ? __builtin_inff() : -__builtin_inff();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      if constexpr (MODE == 0) {          // in-lane CE: v_min + v_max (the pairing alternates, or a repeated CE folds away)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int i = (r + (rep & 1)) & 15, j = (r + 1 + (rep & 1)) & 15;
          float a = fminf(x[i], x[j]), b = fmaxf(x[i], x[j]); x[i] = a; x[j] = b;
        }
      } else if constexpr (MODE == 1) {   // quad_perm dpp + med3
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_fmed3f(x[r], lane_xor<1>(x[r]), c);
      } else if constexpr (MODE == 2) {   // row_mirror dpp + med3
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_fmed3f(x[r], lane_mirror<16>(x[r], lane), c);
      } else if constexpr (MODE == 3) {   // xor4: two bank-masked dpp + med3
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_fmed3f(x[r], lane_xor<4>(x[r]), c);
      } else if constexpr (MODE == 4) {   // med3 only
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_fmed3f(x[r], x[(r + 1) & 15], c);
      } else if constexpr (MODE == 5) {   // v_min with DPP operand folded (VOP2 dpp)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = fminf(x[r], lane_xor<2>(x[r]));
      } else if constexpr (MODE == 6) {   // ds_swizzle + med3
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = __builtin_amdgcn_fmed3f(x[r], lane_xor<16>(x[r]), c);
      } else if constexpr (MODE == 7) {   // integer baseline: v_mad_u32_u24 (independent v_add_f32 pairs become v_pk_add_f32)
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = __int_as_float(__umul24(__float_as_int(x[r]), 3) + lane);
      } else if constexpr (MODE == 9) {   // packed int16 CE between registers: v_pk_min_i16 + v_pk_max_i16 (round 3)
        unsigned* u = reinterpret_cast<unsigned*>(x);
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int i = (r + (rep & 1)) & 15, j = (r + 1 + (rep & 1)) & 15;
          if (rep & 2) pk_ce<true>(u[i], u[j]); else pk_ce<false>(u[i], u[j]);
        }
      } else if constexpr (MODE == 10) {  // the two keys of one register: v_min_i16_sdwa + v_max_i16_sdwa
        unsigned* u = reinterpret_cast<unsigned*>(x);
#pragma unroll
        for (int r = 0; r < 16; ++r) { pk_ce_within(u[r]); u[r] += (unsigned)rep; }
      } else if constexpr (MODE == 11) {  // cross-lane packed stage: 8 DPP moves + 8 v_pk_min under EXEC + 8 v_pk_max under ~EXEC
        unsigned (&u)[8] = *reinterpret_cast<unsigned (*)[8]>(x + 8 * (rep & 1));
        const unsigned t0 = pk_lane_xor<2>(u[0]), t1 = pk_lane_xor<2>(u[1]), t2 = pk_lane_xor<2>(u[2]), t3 = pk_lane_xor<2>(u[3]);
        const unsigned t4 = pk_lane_xor<2>(u[4]), t5 = pk_lane_xor<2>(u[5]), t6 = pk_lane_xor<2>(u[6]), t7 = pk_lane_xor<2>(u[7]);
        pk_lane_stage<0>(u, t0, t1, t2, t3, t4, t5, t6, t7, 0x3333333333333333ull);
      } else if constexpr (MODE == 8) {   // v_cndmask pairs: cmp + 2 cndmask
#pragma unroll
        for (int r = 0; r < 16; r += 2) { bool g = x[r] > x[r + 1]; float a = g ? x[r + 1] : x[r], b = g ? x[r] : x[r + 1]; x[r] = a; x[r + 1] = b; }
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += x[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int valu_per_block, float* d) {
  const int iters = 2000, blocks = 256 * 8;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  // per SIMD: 8 waves x iters x 4 reps x valu_per_block instructions
  double insts = 8.0 * iters * 4 * valu_per_block;
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-34s %7.3f ms  %6.2f cycles per VALU instruction per SIMD (at 2.4 GHz)\n", name, ms, cyc / insts);
}

int main(int argc, char** argv) {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  const int only = argc > 1 ? atoi(argv[1]) : -1;                 // run one mode only (new modes: under `timeout`)
  if (only < 0 || only == 7) run<7>("v_mad_u32_u24", 16, d);
  if (only < 0 || only == 0) run<0>("in-lane CE (v_min+v_max)", 16, d);
  if (only < 0 || only == 4) run<4>("v_med3_f32", 16, d);
  if (only < 0 || only == 1) run<1>("dpp quad_perm mov + med3", 32, d);
  if (only < 0 || only == 2) run<2>("dpp row_mirror mov + med3", 32, d);
  if (only < 0 || only == 3) run<3>("xor4 (2 dpp mov) + med3", 48, d);
  if (only < 0 || only == 5) run<5>("v_min_f32 with dpp operand", 16, d);
  if (only < 0 || only == 6) run<6>("ds_swizzle + med3 (1 VALU)", 16, d);
  if (only < 0 || only == 8) run<8>("cmp + 2 cndmask CE", 24, d);
  if (only < 0 || only == 9) run<9>("packed i16 CE (v_pk_min+v_pk_max)", 16, d);
  if (only < 0 || only == 10) run<10>("within-register CE (2 sdwa) + add", 48, d);
  if (only < 0 || only == 11) run<11>("packed cross-lane stage (8 dpp+16 pk)", 24, d);
  return 0;
}

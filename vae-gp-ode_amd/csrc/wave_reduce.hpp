// wave_reduce.hpp -- sum NV per-lane values over the 64 lanes of a wavefront, result in every lane.
//
// Transposing reduction: instead of NV independent 6-step butterflies (6 NV cross-lane moves + adds),
// pairs of values are folded together level by level, so the number of live registers halves while the
// span of lanes still to be summed halves too:
//   level 1  v_permlane32_swap + add : (a,b) -> [a_lo+a_hi | b_lo+b_hi]            (NV/2 registers)
//   level 2  v_permlane16_swap + add : rows become [v0 v2 v1 v3]                    (NV/4 registers)
//   level 3  4 DPP row_ror adds      : every lane of a 16-lane row holds the row sum
//   finish   v_readlane from lanes 0/16/32/48 -> wave-uniform results
// 8 values cost 4+4+2+2+8 = 20 VALU ops + 8 readlanes (a plain butterfly: 96 ops through LDS).
#pragma once
#include <hip/hip_runtime.h>

namespace gp {


template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}

// every lane <- sum over its 16-lane row
__device__ __forceinline__ float row_allreduce16(float x) {
  x += dpp_mov<0x128>(x);  // row_ror:8
  x += dpp_mov<0x124>(x);  // row_ror:4
  x += dpp_mov<0x122>(x);  // row_ror:2
  x += dpp_mov<0x121>(x);  // row_ror:1
  return x;
}

// [a_lo + a_hi | b_lo + b_hi]: lanes 0..31 carry a's half-sums, lanes 32..63 carry b's
__device__ __forceinline__ float fold32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// rows of p: [x0 x0 x1 x1], rows of q: [y0 y0 y1 y1] (two row-partials each) -> rows [x0 y0 x1 y1]
__device__ __forceinline__ float fold16(float p, float q) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(q), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#define GP_LANE(v, l) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (l)))

// out[i] = sum over the 64 lanes of v[i]; out is wave-uniform.  NV <= 8.
template <int NV> __device__ __forceinline__ void wave_sum_multi(const float (&v)[NV], float (&out)[NV]) {
  static_assert(NV >= 1 && NV <= 8, "1..8 values per call");
  float w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = i < NV ? v[i] : 0.f;
  if constexpr (NV <= 2) {
    float p = row_allreduce16(fold32(w[0], w[1]));  // rows [v0 v0 v1 v1]: add the two rows of each half
    float r0 = GP_LANE(p, 0) + GP_LANE(p, 16);
    out[0] = r0;
    if constexpr (NV == 2) out[1] = GP_LANE(p, 32) + GP_LANE(p, 48);
  } else if constexpr (NV <= 4) {
    float q = row_allreduce16(fold16(fold32(w[0], w[1]), fold32(w[2], w[3])));  // rows [v0 v2 v1 v3]
    out[0] = GP_LANE(q, 0);
    out[1] = GP_LANE(q, 32);
    out[2] = GP_LANE(q, 16);
    if constexpr (NV == 4) out[3] = GP_LANE(q, 48);
  } else {
    float q0 = row_allreduce16(fold16(fold32(w[0], w[1]), fold32(w[2], w[3])));
    float q1 = row_allreduce16(fold16(fold32(w[4], w[5]), fold32(w[6], w[7])));
    out[0] = GP_LANE(q0, 0); out[1] = GP_LANE(q0, 32); out[2] = GP_LANE(q0, 16); out[3] = GP_LANE(q0, 48);
    out[4] = GP_LANE(q1, 0);
    if constexpr (NV > 5) out[5] = GP_LANE(q1, 32);
    if constexpr (NV > 6) out[6] = GP_LANE(q1, 16);
    if constexpr (NV > 7) out[7] = GP_LANE(q1, 48);
  }
}

// any NV: chunks of 8
template <int NV> __device__ __forceinline__ void wave_sum_all(const float (&v)[NV], float (&out)[NV]) {
  if constexpr (NV <= 8) {
    wave_sum_multi<NV>(v, out);
  } else {
    float a[8], ao[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = v[i];
    wave_sum_multi<8>(a, ao);
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = ao[i];
    float b[NV - 8], bo[NV - 8];
#pragma unroll
    for (int i = 0; i < NV - 8; ++i) b[i] = v[8 + i];
    wave_sum_all<NV - 8>(b, bo);
#pragma unroll
    for (int i = 0; i < NV - 8; ++i) out[8 + i] = bo[i];
  }
}

}  // namespace gp

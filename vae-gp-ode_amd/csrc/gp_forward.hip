// gp_forward.hip -- f(x) evaluation and the persistent fixed-grid rollout (forward).
//
// Replaces, per MC draw: SVGP_Layer.forward (svpy.py:123-142) and the whole torchdiffeq solver loop
// behind Flow.forward (flow.py:68-86): T-1 steps x {1|4} RHS evaluations collapse into ONE launch in
// which each wavefront carries one trajectory from z0 to z_{T-1}.
//
// Variants
//   RBF "reg"    : the wave's slice of the packed cache (omega/phase/weights, Z, nu) is loaded once
//                  into VGPRs (cfg1: 216 floats per lane) and reused for all 4(T-1) evaluations.
//   RBF "stream" : any S, M: records are re-read from L2 each evaluation.
//   DF  "lds"    : the pack (cfg2: 98 KB) is staged once per workgroup in LDS, read with ds_read_b128.
//   DF  "stream" : pack larger than LDS (cfg5): records are re-read from L2.
#include "gp_eval.hpp"
#include "gp_team.hpp"
#include "gp_wide.hpp"
#include "gp_launch.hpp"

namespace gp {

// ----------------------------------------------------------------------------------------------
// evaluators (policy objects): operator()(x, f) leaves f(x) in every lane
// ----------------------------------------------------------------------------------------------
template <int DI, int DO, int SJ, int MJ> struct RbfRegEval {
  using L = RbfLayout<DI, DO>;
  float4 rff[SJ * DO][L::RQ];
  float4 ind[MJ][L::RQ2];
  const float* wl;
  __device__ __forceinline__ void init(const float* pack, int M, int S, int lane) {
    const float4* p4 = reinterpret_cast<const float4*>(pack);
#pragma unroll
    for (int r = 0; r < SJ * DO; ++r)
#pragma unroll
      for (int q = 0; q < L::RQ; ++q) rff[r][q] = p4[(r * L::RQ + q) * 64 + lane];
    const float4* i4 = p4 + L::rff_f4(S);
#pragma unroll
    for (int j = 0; j < MJ; ++j)
#pragma unroll
      for (int q = 0; q < L::RQ2; ++q) ind[j][q] = i4[(j * L::RQ2 + q) * 64 + lane];
    wl = pack + 4 * (L::rff_f4(S) + L::ind_f4(M));
  }
  template <int MODE> __device__ __forceinline__ void eval(const float (&x)[DI], float (&f)[DO]) const {
    float acc[DO];
#pragma unroll
    for (int d = 0; d < DO; ++d) acc[d] = 0.f;
    if (MODE != 2) {
#pragma unroll
      for (int j = 0; j < SJ; ++j)
#pragma unroll
        for (int d = 0; d < DO; ++d) rbf_rff_record<DI, DO>(rff[j * DO + d], x, acc[d]);
    }
    if (MODE != 1) {
#pragma unroll
      for (int j = 0; j < MJ; ++j) rbf_ind_record<DI, DO>(ind[j], x, wl, acc);
    }
    wave_sum_all<DO>(acc, f);
  }
};

template <int DI, int DO> struct RbfStreamEval {
  using L = RbfLayout<DI, DO>;
  const float4* rff4;
  const float4* ind4;
  const float* wl;
  int SJ, MJ, lane;
  __device__ __forceinline__ void init(const float* pack, int M, int S, int lane_) {
    rff4 = reinterpret_cast<const float4*>(pack);
    ind4 = rff4 + L::rff_f4(S);
    wl = pack + 4 * (L::rff_f4(S) + L::ind_f4(M));
    SJ = cdiv(S, 64); MJ = cdiv(M, 64); lane = lane_;
  }
  template <int MODE> __device__ __forceinline__ void eval(const float (&x)[DI], float (&f)[DO]) const {
    float acc[DO];
#pragma unroll
    for (int d = 0; d < DO; ++d) acc[d] = 0.f;
    if (MODE != 2) {
      for (int j = 0; j < SJ; ++j) {
#pragma unroll
        for (int d = 0; d < DO; ++d) {
          float4 r[L::RQ];
#pragma unroll
          for (int q = 0; q < L::RQ; ++q) r[q] = rff4[((j * DO + d) * L::RQ + q) * 64 + lane];
          rbf_rff_record<DI, DO>(r, x, acc[d]);
        }
      }
    }
    if (MODE != 1) {
      for (int j = 0; j < MJ; ++j) {
        float4 r[L::RQ2];
#pragma unroll
        for (int q = 0; q < L::RQ2; ++q) r[q] = ind4[(j * L::RQ2 + q) * 64 + lane];
        rbf_ind_record<DI, DO>(r, x, wl, acc);
      }
    }
    wave_sum_all<DO>(acc, f);
  }
};

extern __shared__ __attribute__((aligned(16))) float4 gp_smem4[];

// DF: records from LDS (USE_LDS) or from global/L2.
template <int D, bool USE_LDS> struct DfEval {
  using L = DfLayout<D>;
  const float4* g4;  // global pack (records)
  const float* uni;  // uniform tail (global; scalar loads)
  int SJ, MJ, lane, ind_off;
  __device__ __forceinline__ void init(const float* pack, int M, int S, int lane_) {
    g4 = reinterpret_cast<const float4*>(pack);
    ind_off = (int)L::rff_f4(S);
    uni = pack + 4 * (L::rff_f4(S) + L::ind_f4(M));
    SJ = cdiv(S, 64); MJ = cdiv(M, 64); lane = lane_;
  }
  __device__ __forceinline__ float4 ld(int idx) const {
    if constexpr (USE_LDS) return gp_smem4[idx];
    else return g4[idx];
  }
  template <int MODE> __device__ __forceinline__ void eval(const float (&x)[D], float (&f)[D]) const {
    float acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    if (MODE != 2) {
      for (int j = 0; j < SJ; ++j) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
          float4 r[L::RQ];
#pragma unroll
          for (int q = 0; q < L::RQ; ++q) r[q] = ld(((j * D + i) * L::RQ + q) * 64 + lane);
          df_rff_record<D>(r, x, acc);
        }
      }
    }
    if (MODE != 1) {
      for (int j = 0; j < MJ; ++j) {
        float4 r[L::RQ2];
#pragma unroll
        for (int q = 0; q < L::RQ2; ++q) r[q] = ld(ind_off + (j * L::RQ2 + q) * 64 + lane);
        df_ind_record<D>(r, x, uni, acc);
      }
    }
    wave_sum_all<D>(acc, f);
  }
};

// cooperative copy of the record part of a pack into LDS (whole workgroup), then barrier
__device__ __forceinline__ void stage_pack_lds(const float* pack, size_t n_f4) {
  const float4* g4 = reinterpret_cast<const float4*>(pack);
  for (size_t i = threadIdx.x; i < n_f4; i += blockDim.x) gp_smem4[i] = g4[i];
  __syncthreads();
}

// ----------------------------------------------------------------------------------------------
// kernels
// ----------------------------------------------------------------------------------------------
// x (N,DI) -> f (N,DO); one wave per row, grid-stride over rows.
template <class EV, int DI, int DO, bool USE_LDS>
__global__ __launch_bounds__(256) void rhs_kernel(const float* __restrict__ pack, int M, int S, size_t lds_f4,
                           const float* __restrict__ x, int N, float* __restrict__ f, int mode, Draws dw) {
  pack += blockIdx.y * dw.pack; x += blockIdx.y * dw.in; f += blockIdx.y * dw.out;      // blockIdx.y = Monte-Carlo draw
  if (USE_LDS) stage_pack_lds(pack, lds_f4);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  EV ev;
  ev.init(pack, M, S, lane);
  for (int n = blockIdx.x * wpb + wave; n < N; n += gridDim.x * wpb) {
    float xv[DI], fv[DO];
#pragma unroll
    for (int i = 0; i < DI; ++i) xv[i] = x[(size_t)n * DI + i];
    if (mode == 0) ev.template eval<0>(xv, fv);
    else if (mode == 1) ev.template eval<1>(xv, fv);
    else ev.template eval<2>(xv, fv);
    if (lane < DO) {
      float v = fv[0];
#pragma unroll
      for (int d = 1; d < DO; ++d) v = (lane == d) ? fv[d] : v;
      f[(size_t)n * DO + lane] = v;
    }
  }
}

// ODE right-hand side on the state (flow.py:27-45): order 1: f(y); order 2: [v ; f(s,v)].
template <class EV, int DI, int DO, int ORDER>
__device__ __forceinline__ void ode_rhs(const EV& ev, const float (&y)[DI], float (&dy)[DI]) {
  float fv[DO];
  ev.template eval<0>(y, fv);
  if (ORDER == 1) {
#pragma unroll
    for (int i = 0; i < DO; ++i) dy[i] = fv[i];
  } else {
#pragma unroll
    for (int i = 0; i < DO; ++i) { dy[i] = y[DO + i]; dy[DO + i] = fv[i]; }
  }
}

template <int DI> __device__ __forceinline__ void store_state(float* __restrict__ dst, const float (&y)[DI], int lane) {
  if (lane < DI) {
    float v = y[0];
#pragma unroll
    for (int i = 1; i < DI; ++i) v = (lane == i) ? y[i] : v;
    dst[lane] = v;
  }
}

// z0 (N,DI), ts (T) -> zt (N,T,DI).  One wave per trajectory, persistent over the T-1 steps.
// Stage algebra follows torchdiffeq's fixed-grid solvers at flow.py:76-85 (3/8-rule rk4).
template <class EV, int DI, int DO, int ORDER, int METHOD, bool USE_LDS>
__global__ __launch_bounds__(256) void rollout_kernel(const float* __restrict__ pack, int M, int S, size_t lds_f4,
                               const float* __restrict__ z0, const float* __restrict__ ts, int N, int T,
                               float* __restrict__ zt, float* __restrict__ xstage, Draws dw) {
  static_assert(DI == ORDER * DO, "state dim = order * D_out");
  constexpr int NS = METHOD == 0 ? 1 : (METHOD == 1 ? 4 : 2);
  pack += blockIdx.y * dw.pack; z0 += blockIdx.y * dw.in; zt += blockIdx.y * dw.out;    // blockIdx.y = Monte-Carlo draw
  if (xstage) xstage += blockIdx.y * dw.out2;
  if (USE_LDS) stage_pack_lds(pack, lds_f4);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
  EV ev;
  ev.init(pack, M, S, lane);
  const float third = (float)(1.0 / 3.0);
  for (int n = blockIdx.x * wpb + wave; n < N; n += gridDim.x * wpb) {
    float y[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) y[i] = z0[(size_t)n * DI + i];
    float* out = zt + (size_t)n * T * DI;
    float* xs_out = xstage ? xstage + (size_t)n * (T - 1) * NS * DI : nullptr;
    store_state<DI>(out, y, lane);
    for (int t = 0; t + 1 < T; ++t) {
      const float dt = ts[t + 1] - ts[t];
      float k1[DI];
      if (xs_out) store_state<DI>(xs_out + (size_t)(t * NS) * DI, y, lane);
      ode_rhs<EV, DI, DO, ORDER>(ev, y, k1);
      if (METHOD == 0) {
#pragma unroll
        for (int i = 0; i < DI; ++i) y[i] = y[i] + dt * k1[i];
      } else if (METHOD == 2) {                      // midpoint: y1 = y + dt f(y + dt/2 f(y))
        float k2[DI], xs[DI];
#pragma unroll
        for (int i = 0; i < DI; ++i) xs[i] = y[i] + 0.5f * dt * k1[i];
        if (xs_out) store_state<DI>(xs_out + (size_t)(t * NS + 1) * DI, xs, lane);
        ode_rhs<EV, DI, DO, ORDER>(ev, xs, k2);
#pragma unroll
        for (int i = 0; i < DI; ++i) y[i] = y[i] + dt * k2[i];
      } else {
        float k2[DI], k3[DI], k4[DI], xs[DI];
#pragma unroll
        for (int i = 0; i < DI; ++i) xs[i] = y[i] + dt * k1[i] * third;
        if (xs_out) store_state<DI>(xs_out + (size_t)(t * NS + 1) * DI, xs, lane);
        ode_rhs<EV, DI, DO, ORDER>(ev, xs, k2);
#pragma unroll
        for (int i = 0; i < DI; ++i) xs[i] = y[i] + dt * (k2[i] - k1[i] * third);
        if (xs_out) store_state<DI>(xs_out + (size_t)(t * NS + 2) * DI, xs, lane);
        ode_rhs<EV, DI, DO, ORDER>(ev, xs, k3);
#pragma unroll
        for (int i = 0; i < DI; ++i) xs[i] = y[i] + dt * (k1[i] - k2[i] + k3[i]);
        if (xs_out) store_state<DI>(xs_out + (size_t)(t * NS + 3) * DI, xs, lane);
        ode_rhs<EV, DI, DO, ORDER>(ev, xs, k4);
#pragma unroll
        for (int i = 0; i < DI; ++i) y[i] = y[i] + (k1[i] + 3.f * (k2[i] + k3[i]) + k4[i]) * dt * 0.125f;
      }
      store_state<DI>(out + (size_t)(t + 1) * DI, y, lane);
    }
  }
}

template <class EV, int DI, int DO>
__global__ __launch_bounds__(64 * EV::kTeam) void rhs_team_kernel(const float* __restrict__ pack, int M, int S,
                                                        const float* __restrict__ x, int N, float* __restrict__ f, int mode, Draws dw) {
  __shared__ float slots[2 * EV::kTeam * TeamCombine::DP];
  pack += blockIdx.y * dw.pack; x += blockIdx.y * dw.in; f += blockIdx.y * dw.out;      // blockIdx.y = Monte-Carlo draw
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  EV ev;
  ev.init(pack, M, S, slots, wave, lane);
  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    float xv[DI], fv[DO];
#pragma unroll
    for (int i = 0; i < DI; ++i) xv[i] = x[(size_t)n * DI + i];
    if (mode == 0) ev.template eval<0>(xv, fv);
    else if (mode == 1) ev.template eval<1>(xv, fv);
    else ev.template eval<2>(xv, fv);
    if (wave == 0 && lane < DO) {
      float v = fv[0];
#pragma unroll
      for (int d = 1; d < DO; ++d) v = (lane == d) ? fv[d] : v;
      f[(size_t)n * DO + lane] = v;
    }
  }
}

template <class EV, int DI, int DO, int ORDER>
__device__ __forceinline__ void ode_rhs_mut(EV& ev, const float (&y)[DI], float (&dy)[DI]) {
  float fv[DO];
  ev.template eval<0>(y, fv);
  if (ORDER == 1) {
#pragma unroll
    for (int i = 0; i < DO; ++i) dy[i] = fv[i];
  } else {
#pragma unroll
    for (int i = 0; i < DO; ++i) { dy[i] = y[DO + i]; dy[DO + i] = fv[i]; }
  }
}

template <class EV, int DI, int DO, int ORDER, int METHOD>
__global__ __launch_bounds__(64 * EV::kTeam) void rollout_team_kernel(const float* __restrict__ pack, int M, int S,
                                                            const float* __restrict__ z0, const float* __restrict__ ts,
                                                            int N, int T, float* __restrict__ zt, float* __restrict__ xstage, Draws dw) {
  static_assert(DI == ORDER * DO, "state dim = order * D_out");
  constexpr int NS = METHOD == 0 ? 1 : (METHOD == 1 ? 4 : 2);
  __shared__ float slots[2 * EV::kTeam * TeamCombine::DP];
  // blockIdx.y = Monte-Carlo draw: its own pack (function draw), the shared initial states, its own trajectories
  pack += blockIdx.y * dw.pack; z0 += blockIdx.y * dw.in; zt += blockIdx.y * dw.out;
  if (xstage) xstage += blockIdx.y * dw.out2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  EV ev;
  ev.init(pack, M, S, slots, wave, lane);
  const float third = (float)(1.0 / 3.0);
  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    float y[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) y[i] = z0[(size_t)n * DI + i];
    float* out = zt + (size_t)n * T * DI;
    float* xs_out = (xstage && wave == 0) ? xstage + (size_t)n * (T - 1) * NS * DI : nullptr;
    if (wave == 0) store_state<DI>(out, y, lane);
    for (int t = 0; t + 1 < T; ++t) {
      const float dt = ts[t + 1] - ts[t];
      float k1[DI];
      if (xs_out) store_state<DI>(xs_out + (size_t)(t * NS) * DI, y, lane);
      ode_rhs_mut<EV, DI, DO, ORDER>(ev, y, k1);
      if (METHOD == 0) {
#pragma unroll
        for (int i = 0; i < DI; ++i) y[i] = y[i] + dt * k1[i];
      } else if (METHOD == 2) {                      // midpoint
        float k2[DI], xs[DI];
#pragma unroll
        for (int i = 0; i < DI; ++i) xs[i] = y[i] + 0.5f * dt * k1[i];
        if (xs_out) store_state<DI>(xs_out + (size_t)(t * NS + 1) * DI, xs, lane);
        ode_rhs_mut<EV, DI, DO, ORDER>(ev, xs, k2);
#pragma unroll
        for (int i = 0; i < DI; ++i) y[i] = y[i] + dt * k2[i];
      } else {
        float k2[DI], k3[DI], k4[DI], xs[DI];
#pragma unroll
        for (int i = 0; i < DI; ++i) xs[i] = y[i] + dt * k1[i] * third;
        if (xs_out) store_state<DI>(xs_out + (size_t)(t * NS + 1) * DI, xs, lane);
        ode_rhs_mut<EV, DI, DO, ORDER>(ev, xs, k2);
#pragma unroll
        for (int i = 0; i < DI; ++i) xs[i] = y[i] + dt * (k2[i] - k1[i] * third);
        if (xs_out) store_state<DI>(xs_out + (size_t)(t * NS + 2) * DI, xs, lane);
        ode_rhs_mut<EV, DI, DO, ORDER>(ev, xs, k3);
#pragma unroll
        for (int i = 0; i < DI; ++i) xs[i] = y[i] + dt * (k1[i] - k2[i] + k3[i]);
        if (xs_out) store_state<DI>(xs_out + (size_t)(t * NS + 3) * DI, xs, lane);
        ode_rhs_mut<EV, DI, DO, ORDER>(ev, xs, k4);
#pragma unroll
        for (int i = 0; i < DI; ++i) y[i] = y[i] + (k1[i] + 3.f * (k2[i] + k3[i]) + k4[i]) * dt * 0.125f;
      }
      if (wave == 0) store_state<DI>(out + (size_t)(t + 1) * DI, y, lane);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// host launchers
// ----------------------------------------------------------------------------------------------
static const size_t kLdsLimitBytes = 150 * 1024;  // 160 KiB per CU; leave headroom

template <int DI, int DO, int SJ, int MJ> constexpr bool rbf_reg_fits() {
  return 4 * (SJ * DO * RbfLayout<DI, DO>::RQ + MJ * RbfLayout<DI, DO>::RQ2) <= 260;
}

static inline void grid_for(int N, int& grid, int& block) {
  // one wave per row.  Few rows: 1 wave per workgroup so they spread over all 256 CUs;
  // many rows: 4 waves per workgroup, at most 2 workgroups per CU resident, grid-stride beyond.
  if (N <= 1024) { block = 64; grid = N; }
  else { block = 256; grid = (N + 3) / 4; if (grid > 2048) grid = 2048; }
  if (grid < 1) grid = 1;
}


// wavefronts per trajectory of the wide team (gp_wide.hpp) for an output width, 0: none
template <int DO> constexpr int wide_ts() { return (DO == 6 || DO == 3) ? 12 : (DO == 4 ? 8 : (DO == 8 ? 16 : 0)); }

static const int kTeamMaxRows = 2048;  // below this, 4 waves per trajectory beat 1 (all 1024 SIMDs busy sooner)
static inline int team_grid(int N) { return N < 2048 ? N : 2048; }

template <int DI, int DO>
static int launch_rhs_rbf(const float* pack, int M, int S, const float* x, int N, float* f, int mode, hipStream_t st, Draws dw) {
  int grid, block;
  grid_for(N, grid, block);
  if (N <= kTeamMaxRows && DO <= 16) {
    if (RbfTeamEval<DI, DO, 1>::fits(M, S)) {
      hipLaunchKernelGGL((rhs_team_kernel<RbfTeamEval<DI, DO, 1>, DI, DO>), dim3(team_grid(N), dw.nd), 256, 0, st, pack, M, S, x, N, f, mode, dw);
      return check_launch("rhs_rbf_team");
    }
  }
  if (N <= kTeamMaxRows) {     // past the register-resident quarter pack: the same team, records streamed from L2
    hipLaunchKernelGGL((rhs_team_kernel<RbfStreamTeam<DI, DO>, DI, DO>), dim3(team_grid(N), dw.nd), 256, 0, st, pack, M, S, x, N, f, mode, dw);
    return check_launch("rhs_rbf_team_stream");
  }
  const int SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  if constexpr (rbf_reg_fits<DI, DO, 4, 2>()) {
    if (SJ == 4 && MJ == 2) {
      hipLaunchKernelGGL((rhs_kernel<RbfRegEval<DI, DO, 4, 2>, DI, DO, false>), dim3(grid, dw.nd), block, 0, st, pack, M, S, (size_t)0, x, N, f, mode, dw);
      return check_launch("rhs_rbf");
    }
  }
  if constexpr (rbf_reg_fits<DI, DO, 1, 1>()) {
    if (SJ == 1 && MJ == 1) {
      hipLaunchKernelGGL((rhs_kernel<RbfRegEval<DI, DO, 1, 1>, DI, DO, false>), dim3(grid, dw.nd), block, 0, st, pack, M, S, (size_t)0, x, N, f, mode, dw);
      return check_launch("rhs_rbf");
    }
  }
  hipLaunchKernelGGL((rhs_kernel<RbfStreamEval<DI, DO>, DI, DO, false>), dim3(grid, dw.nd), block, 0, st, pack, M, S, (size_t)0, x, N, f, mode, dw);
  return check_launch("rhs_rbf");
}

template <int D>
static int launch_rhs_df(const float* pack, int M, int S, const float* x, int N, float* f, int mode, hipStream_t st, Draws dw) {
  using L = DfLayout<D>;
  const size_t f4 = L::rff_f4(S) + L::ind_f4(M);
  int grid, block;
  grid_for(N, grid, block);
  if constexpr (D <= 8) {
    if (N <= kTeamMaxRows && DfTeamEval<D, 1>::fits(M, S)) {
      hipLaunchKernelGGL((rhs_team_kernel<DfTeamEval<D, 1>, D, D>), dim3(team_grid(N), dw.nd), 256, 0, st, pack, M, S, x, N, f, mode, dw);
      return check_launch("rhs_df_team");
    }
  }
  if (N <= kTeamMaxRows) {
    hipLaunchKernelGGL((rhs_team_kernel<DfStreamTeam<D>, D, D>), dim3(team_grid(N), dw.nd), 256, 0, st, pack, M, S, x, N, f, mode, dw);
    return check_launch("rhs_df_team_stream");
  }
  // one evaluation per row: staging the pack in LDS only pays when a workgroup evaluates many rows
  if (f4 * 16 <= kLdsLimitBytes && N >= 2048) {
    block = 256; grid = 256;
    auto kern = rhs_kernel<DfEval<D, true>, D, D, true>;
    if (set_max_lds((const void*)kern, f4 * 16)) return 1;
    hipLaunchKernelGGL(kern, dim3(grid, dw.nd), block, f4 * 16, st, pack, M, S, f4, x, N, f, mode, dw);
  } else {
    hipLaunchKernelGGL((rhs_kernel<DfEval<D, false>, D, D, false>), dim3(grid, dw.nd), block, 0, st, pack, M, S, (size_t)0, x, N, f, mode, dw);
  }
  return check_launch("rhs_df");
}

template <int DI, int DO, int ORDER, int METHOD>
static int launch_rollout_rbf(const float* pack, int M, int S, const float* z0, const float* ts, int N, int T, float* zt, float* xstage, hipStream_t st, Draws dw) {
  int grid, block;
  grid_for(N, grid, block);
  if constexpr (wide_ts<DO>() > 0 && DI <= 8) {
    constexpr int TS = wide_ts<DO>();
    if (wide_team_enabled() && N <= kWideMaxRows && RbfWideTeam<DI, DO, TS>::fits(M, S)) {
      hipLaunchKernelGGL((rollout_team_kernel<RbfWideTeam<DI, DO, TS>, DI, DO, ORDER, METHOD>), dim3(N, dw.nd), 64 * TS, 0, st, pack, M, S, z0, ts, N, T, zt, xstage, dw);
      return check_launch("rollout_rbf_wide");
    }
  }
  if (N <= kTeamMaxRows && DO <= 16) {
    if (RbfTeamEval<DI, DO, 1>::fits(M, S)) {
      hipLaunchKernelGGL((rollout_team_kernel<RbfTeamEval<DI, DO, 1>, DI, DO, ORDER, METHOD>), dim3(team_grid(N), dw.nd), 256, 0, st, pack, M, S, z0, ts, N, T, zt, xstage, dw);
      return check_launch("rollout_rbf_team");
    }
  }
  if (N <= kTeamMaxRows) {
    hipLaunchKernelGGL((rollout_team_kernel<RbfStreamTeam<DI, DO>, DI, DO, ORDER, METHOD>), dim3(team_grid(N), dw.nd), 256, 0, st, pack, M, S, z0, ts, N, T, zt, xstage, dw);
    return check_launch("rollout_rbf_team_stream");
  }
  const int SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  if constexpr (rbf_reg_fits<DI, DO, 4, 2>()) {
    if (SJ == 4 && MJ == 2) {
      hipLaunchKernelGGL((rollout_kernel<RbfRegEval<DI, DO, 4, 2>, DI, DO, ORDER, METHOD, false>), dim3(grid, dw.nd), block, 0, st, pack, M, S, (size_t)0, z0, ts, N, T, zt, xstage, dw);
      return check_launch("rollout_rbf");
    }
  }
  if constexpr (rbf_reg_fits<DI, DO, 1, 1>()) {
    if (SJ == 1 && MJ == 1) {
      hipLaunchKernelGGL((rollout_kernel<RbfRegEval<DI, DO, 1, 1>, DI, DO, ORDER, METHOD, false>), dim3(grid, dw.nd), block, 0, st, pack, M, S, (size_t)0, z0, ts, N, T, zt, xstage, dw);
      return check_launch("rollout_rbf");
    }
  }
  hipLaunchKernelGGL((rollout_kernel<RbfStreamEval<DI, DO>, DI, DO, ORDER, METHOD, false>), dim3(grid, dw.nd), block, 0, st, pack, M, S, (size_t)0, z0, ts, N, T, zt, xstage, dw);
  return check_launch("rollout_rbf");
}

template <int D, int METHOD>
static int launch_rollout_df(const float* pack, int M, int S, const float* z0, const float* ts, int N, int T, float* zt, float* xstage, hipStream_t st, Draws dw) {
  using L = DfLayout<D>;
  const size_t f4 = L::rff_f4(S) + L::ind_f4(M);
  int grid, block;
  grid_for(N, grid, block);
  if constexpr (wide_ts<D>() > 0) {
    constexpr int TS = wide_ts<D>();
    if (wide_team_enabled() && N <= kWideMaxRows && DfWideTeam<D, TS>::fits(M, S)) {
      hipLaunchKernelGGL((rollout_team_kernel<DfWideTeam<D, TS>, D, D, 1, METHOD>), dim3(N, dw.nd), 64 * TS, 0, st, pack, M, S, z0, ts, N, T, zt, xstage, dw);
      return check_launch("rollout_df_wide");
    }
  }
  if constexpr (D <= 8) {
    if (N <= kTeamMaxRows && DfTeamEval<D, 1>::fits(M, S)) {
      hipLaunchKernelGGL((rollout_team_kernel<DfTeamEval<D, 1>, D, D, 1, METHOD>), dim3(team_grid(N), dw.nd), 256, 0, st, pack, M, S, z0, ts, N, T, zt, xstage, dw);
      return check_launch("rollout_df_team");
    }
  }
  if (N <= kTeamMaxRows) {     // e.g. BASELINE configs[4] (D = 16, M = 512): 4 wavefronts per trajectory, records streamed from L2
    hipLaunchKernelGGL((rollout_team_kernel<DfStreamTeam<D>, D, D, 1, METHOD>), dim3(team_grid(N), dw.nd), 256, 0, st, pack, M, S, z0, ts, N, T, zt, xstage, dw);
    return check_launch("rollout_df_team_stream");
  }
  if (f4 * 16 <= kLdsLimitBytes) {
    if (N <= 1024) { block = 64; grid = N < 256 ? N : 256; }
    else { block = 256; grid = 256; }
    auto kern = rollout_kernel<DfEval<D, true>, D, D, 1, METHOD, true>;
    if (set_max_lds((const void*)kern, f4 * 16)) return 1;
    hipLaunchKernelGGL(kern, dim3(grid, dw.nd), block, f4 * 16, st, pack, M, S, f4, z0, ts, N, T, zt, xstage, dw);
  } else {
    hipLaunchKernelGGL((rollout_kernel<DfEval<D, false>, D, D, 1, METHOD, false>), dim3(grid, dw.nd), block, 0, st, pack, M, S, (size_t)0, z0, ts, N, T, zt, xstage, dw);
  }
  return check_launch("rollout_df");
}

// dispatch tables -----------------------------------------------------------------------------
#define GP_RBF_DIMS(X) X(6, 6) X(6, 3) X(4, 4) X(4, 2) X(2, 2) X(2, 1) X(8, 8) X(8, 4) X(16, 16) X(16, 8) X(3, 3) X(12, 6)
#define GP_DF_DIMS(X) X(6) X(4) X(2) X(3) X(8) X(16) X(5) X(7) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

int rhs_fwd(int kernel, int Di, int Do, int M, int S, const float* pack, const float* x, int N, float* f, int mode, hipStream_t st, Draws dw) {
  if (kernel == 0) {
#define X(a, b) if (Di == a && Do == b) return launch_rhs_rbf<a, b>(pack, M, S, x, N, f, mode, st, dw);
    GP_RBF_DIMS(X)
#undef X
  } else {
#define X(a) if (Di == a && Do == a) return launch_rhs_df<a>(pack, M, S, x, N, f, mode, st, dw);
    GP_DF_DIMS(X)
#undef X
  }
  return set_error("gpode_rhs_fwd: no specialisation for kernel=%d Di=%d Do=%d", kernel, Di, Do);
}

template <int DI, int DO>
static int rollout_rbf_dispatch(int order, int method, const float* pack, int M, int S, const float* z0, const float* ts, int N, int T, float* zt, float* xstage, hipStream_t st, Draws dw) {
  if constexpr (DI == DO) {
    if (order == 1 && method == 0) return launch_rollout_rbf<DI, DO, 1, 0>(pack, M, S, z0, ts, N, T, zt, xstage, st, dw);
    if (order == 1 && method == 1) return launch_rollout_rbf<DI, DO, 1, 1>(pack, M, S, z0, ts, N, T, zt, xstage, st, dw);
    if (order == 1 && method == 2) return launch_rollout_rbf<DI, DO, 1, 2>(pack, M, S, z0, ts, N, T, zt, xstage, st, dw);
  }
  if constexpr (DI == 2 * DO) {
    if (order == 2 && method == 0) return launch_rollout_rbf<DI, DO, 2, 0>(pack, M, S, z0, ts, N, T, zt, xstage, st, dw);
    if (order == 2 && method == 1) return launch_rollout_rbf<DI, DO, 2, 1>(pack, M, S, z0, ts, N, T, zt, xstage, st, dw);
    if (order == 2 && method == 2) return launch_rollout_rbf<DI, DO, 2, 2>(pack, M, S, z0, ts, N, T, zt, xstage, st, dw);
  }
  return set_error("gpode_rollout_fwd: order=%d needs Di == order*Do (Di=%d Do=%d)", order, DI, DO);
}

int rollout_fwd(int kernel, int order, int method, int Di, int Do, int M, int S, const float* pack,
                const float* z0, const float* ts, int N, int T, float* zt, float* xstage, hipStream_t st, Draws dw) {
  if (method < 0 || method > 2) return set_error("gpode_rollout_fwd: method %d (0 euler, 1 rk4, 2 midpoint)", method);
  if (kernel == 0) {
#define X(a, b) if (Di == a && Do == b) return rollout_rbf_dispatch<a, b>(order, method, pack, M, S, z0, ts, N, T, zt, xstage, st, dw);
    GP_RBF_DIMS(X)
#undef X
  } else {
    if (order != 1) return set_error("gpode_rollout_fwd: DF kernel is first-order only (kernels.py:259-262)");
#define X(a) if (Di == a && Do == a) return method == 0 ? launch_rollout_df<a, 0>(pack, M, S, z0, ts, N, T, zt, xstage, st, dw) \
                                            : method == 1 ? launch_rollout_df<a, 1>(pack, M, S, z0, ts, N, T, zt, xstage, st, dw) \
                                                          : launch_rollout_df<a, 2>(pack, M, S, z0, ts, N, T, zt, xstage, st, dw);
    GP_DF_DIMS(X)
#undef X
  }
  return set_error("gpode_rollout_fwd: no specialisation for kernel=%d Di=%d Do=%d", kernel, Di, Do);
}

int dims_supported(int kernel, int Di, int Do) {
  if (kernel == 0) {
#define X(a, b) if (Di == a && Do == b) return 1;
    GP_RBF_DIMS(X)
#undef X
  } else if (kernel == 1) {
#define X(a) if (Di == a && Do == a) return 1;
    GP_DF_DIMS(X)
#undef X
  }
  return 0;
}

}  // namespace gp

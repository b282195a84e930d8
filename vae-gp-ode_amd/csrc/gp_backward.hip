// gp_backward.hip -- reverse sweep of the fixed-grid rollout and parameter gradients of f.
//
// The reference back-propagates through the unrolled solver with autograd (use_adjoint=False,
// main.py:85,209-210): 4(T-1) RHS graphs per draw.  Here the same discretise-then-optimise gradient is
// computed by two kernels per draw:
//
//  A  rollout_bwd_team_kernel  one workgroup (4 waves) per trajectory walks the steps backwards over the
//     stage inputs the forward stored (xstage), applying the adjoint of the 3/8-rule (or Euler) stage
//     algebra; each stage costs one vector-Jacobian product J_f(x)^T a.  Outputs dL/dz0 and the stage
//     adjoints a (astage).  Sequential in t, parallel over trajectories.
//  B  param_grad_team_kernel   with (x, a) known for every (trajectory, step, stage) row, the parameter
//     gradient is a plain sum over rows -- no sequential dependence left.  Each wave keeps its quarter of
//     the pack AND the matching gradient accumulators in registers and streams a chunk of rows; chunk
//     results go to a slab that a second kernel sums in fixed order (deterministic, no float atomics).
//     Gradients come out in PACK LAYOUT (d/d om, d/d aw, d/d z, d/d cc ... + the uniform tail), which
//     gp_cache_bwd.hip chains back to the raw parameters.
#include "gp_eval.hpp"
#include "gp_team.hpp"
#include "gp_wide.hpp"
#include "gp_launch.hpp"

namespace gp {

template <int DI> __device__ __forceinline__ void store_vec(float* __restrict__ dst, const float (&y)[DI], int lane) {
  if (lane < DI) {
    float v = y[0];
#pragma unroll
    for (int i = 1; i < DI; ++i) v = (lane == i) ? y[i] : v;
    dst[lane] = v;
  }
}

// ODE-level VJP (flow.py:27-45): order 1: dy = f(y); order 2: dy = [y[q:], f(y)].
// a (DI) = adjoint of dy  ->  gx (DI) = (d dy / d y)^T a ;  af (DO) = the part that multiplies J_f.
template <int DI, int DO, int NJ>
__device__ __forceinline__ void store_grads_rbf(const typename RbfTeamEval<DI, DO, NJ>::Grads& G, float* __restrict__ out, int M, int S,
                                                int wave, int lane, float (*sInd)[64][4 * RbfLayout<DI, DO>::RQ2],
                                                float (*sUni)[((DO + 1) / 2) * DI]);
template <int D, int NJ, int PART = 0>
__device__ __forceinline__ void store_grads_df(const typename DfTeamEval<D, NJ>::Grads& G, float* __restrict__ out, int M, int S, int wave,
                                               int lane, float (*sInd)[64][4 * DfLayout<D>::RQ2],
                                               float (*sUni)[2 * D * ((D + 1) / 2) + (D + 1) / 2]);
struct NoGrads {};
template <class EV, int DI, int DO, int ORDER, class GR = NoGrads>
__device__ __forceinline__ void ode_vjp(EV& ev, const float (&x)[DI], const float (&a)[DI], float (&gx)[DI], float (&af)[DO], GR* G = nullptr) {
#pragma unroll
  for (int i = 0; i < DO; ++i) af[i] = ORDER == 1 ? a[i] : a[DO + i];
  if constexpr (std::is_same<GR, NoGrads>::value) ev.vjp(x, af, gx);
  else ev.vjp_grad(x, af, gx, *G);                  // the row's parameter-gradient terms ride along (PGRAD form of the reverse sweep)
  if (ORDER != 1) {
#pragma unroll
    for (int i = 0; i < DO; ++i) gx[DO + i] += a[i];
  }
}

template <class EV, bool PG> struct GradsOf { using type = NoGrads; };
template <class EV> struct GradsOf<EV, true> { using type = typename EV::Grads; };

// PG (register-resident team evaluators only): the parameter-gradient sums of the rows this workgroup walks are accumulated in
// the same pass (EV::vjp_grad) and written to slab[blockIdx.x] in pack layout -- what param_grad_{rbf,df}_kernel computes from the
// stored (x, a) rows in a launch of its own, the first node of the backward pass's side branch.
template <class EV, int DI, int DO, int ORDER, int METHOD, bool PG = false>
__global__ __launch_bounds__(64 * EV::kTeam) void rollout_bwd_team_kernel(const float* __restrict__ pack, int M, int S,
                                                                const float* __restrict__ xstage, const float* __restrict__ gzt,
                                                                const float* __restrict__ ts, int N, int T,
                                                                float* __restrict__ gz0, float* __restrict__ astage, Draws dw,
                                                                float* __restrict__ slab = nullptr, size_t pack_floats = 0) {
  constexpr int NS = METHOD == 0 ? 1 : (METHOD == 1 ? 4 : 2);
  __shared__ float slots[2 * EV::kTeam * TeamCombine::DP];
  // blockIdx.y = Monte-Carlo draw
  pack += blockIdx.y * dw.pack; xstage += blockIdx.y * dw.in; gzt += blockIdx.y * dw.in2; gz0 += blockIdx.y * dw.out; astage += blockIdx.y * dw.out2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  EV ev;
  ev.init(pack, M, S, slots, wave, lane);
  using GR = typename GradsOf<EV, PG>::type;
  GR G;
  if constexpr (PG) G.zero();
  const float third = (float)(1.0 / 3.0);
  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    const float* gz = gzt + (size_t)n * T * DI;
    const float* xs = xstage + (size_t)n * (T - 1) * NS * DI;
    float* as = astage + (size_t)n * (T - 1) * NS * DO;
    float lam[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) lam[i] = gz[(size_t)(T - 1) * DI + i];
    for (int t = T - 2; t >= 0; --t) {
      const float dt = ts[t + 1] - ts[t];
      const float* xt = xs + (size_t)t * NS * DI;
      float* at = as + (size_t)t * NS * DO;
      float x[DI], g[DI], af[DO];
      if (METHOD == 0) {
        float a1[DI];
#pragma unroll
        for (int i = 0; i < DI; ++i) { a1[i] = dt * lam[i]; x[i] = xt[i]; }
        ode_vjp<EV, DI, DO, ORDER, GR>(ev, x, a1, g, af, &G);
        if (wave == 0) store_vec<DO>(at, af, lane);
#pragma unroll
        for (int i = 0; i < DI; ++i) lam[i] += g[i];
      } else if (METHOD == 2) {
        // midpoint: y1 = y + dt k2, k2 = f(x2), x2 = y + dt/2 k1, k1 = f(y)
        float a1[DI], a2[DI], ay[DI];
#pragma unroll
        for (int i = 0; i < DI; ++i) { a2[i] = dt * lam[i]; x[i] = xt[DI + i]; }
        ode_vjp<EV, DI, DO, ORDER, GR>(ev, x, a2, g, af, &G);
        if (wave == 0) store_vec<DO>(at + DO, af, lane);
#pragma unroll
        for (int i = 0; i < DI; ++i) { ay[i] = lam[i] + g[i]; a1[i] = 0.5f * dt * g[i]; x[i] = xt[i]; }
        ode_vjp<EV, DI, DO, ORDER, GR>(ev, x, a1, g, af, &G);
        if (wave == 0) store_vec<DO>(at, af, lane);
#pragma unroll
        for (int i = 0; i < DI; ++i) lam[i] = ay[i] + g[i];
      } else {
        // y1 = y + (k1 + 3(k2+k3) + k4) dt/8 ;  x2 = y + dt k1/3 ; x3 = y + dt(k2 - k1/3) ; x4 = y + dt(k1 - k2 + k3)
        float a1[DI], a2[DI], a3[DI], a4[DI], ay[DI];
#pragma unroll
        for (int i = 0; i < DI; ++i) {
          const float l8 = lam[i] * dt * 0.125f;
          a1[i] = l8; a4[i] = l8; a2[i] = 3.f * l8; a3[i] = 3.f * l8; ay[i] = lam[i];
        }
#pragma unroll
        for (int i = 0; i < DI; ++i) x[i] = xt[3 * DI + i];
        ode_vjp<EV, DI, DO, ORDER, GR>(ev, x, a4, g, af, &G);
        if (wave == 0) store_vec<DO>(at + 3 * DO, af, lane);
#pragma unroll
        for (int i = 0; i < DI; ++i) { ay[i] += g[i]; a1[i] += dt * g[i]; a2[i] -= dt * g[i]; a3[i] += dt * g[i]; }
#pragma unroll
        for (int i = 0; i < DI; ++i) x[i] = xt[2 * DI + i];
        ode_vjp<EV, DI, DO, ORDER, GR>(ev, x, a3, g, af, &G);
        if (wave == 0) store_vec<DO>(at + 2 * DO, af, lane);
#pragma unroll
        for (int i = 0; i < DI; ++i) { ay[i] += g[i]; a2[i] += dt * g[i]; a1[i] -= dt * third * g[i]; }
#pragma unroll
        for (int i = 0; i < DI; ++i) x[i] = xt[1 * DI + i];
        ode_vjp<EV, DI, DO, ORDER, GR>(ev, x, a2, g, af, &G);
        if (wave == 0) store_vec<DO>(at + 1 * DO, af, lane);
#pragma unroll
        for (int i = 0; i < DI; ++i) { ay[i] += g[i]; a1[i] += dt * third * g[i]; }
#pragma unroll
        for (int i = 0; i < DI; ++i) x[i] = xt[i];
        ode_vjp<EV, DI, DO, ORDER, GR>(ev, x, a1, g, af, &G);
        if (wave == 0) store_vec<DO>(at, af, lane);
#pragma unroll
        for (int i = 0; i < DI; ++i) lam[i] = ay[i] + g[i];
      }
#pragma unroll
      for (int i = 0; i < DI; ++i) lam[i] += gz[(size_t)t * DI + i];
    }
    if (wave == 0) store_vec<DI>(gz0 + (size_t)n * DI, lam, lane);
  }
  if constexpr (PG) {
    float* out = slab + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * pack_floats;
    if constexpr (std::is_same<EV, RbfTeamEval<DI, DO, 1>>::value) {
      using L = RbfLayout<DI, DO>;
      __shared__ __attribute__((aligned(16))) float sInd[2][64][4 * L::RQ2];
      __shared__ float sUni[TEAM][((DO + 1) / 2) * DI];
      store_grads_rbf<DI, DO, 1>(G, out, M, S, wave, lane, sInd, sUni);
    } else {
      using L = DfLayout<DO>;
      constexpr int DH = (DO + 1) / 2;
      __shared__ __attribute__((aligned(16))) float sInd[2][64][4 * L::RQ2];
      __shared__ float sUni[TEAM][2 * DO * DH + DH];
      store_grads_df<DO, 1>(G, out, M, S, wave, lane, sInd, sUni);
    }
  }
}

// rows (R,DI), adjoints (R,DO) -> gx (R,DI) = J_f(x)^T a   (used for the f_prior(Z) path of the cache backward)
template <class EV, int DI, int DO>
__global__ __launch_bounds__(64 * EV::kTeam) void rhs_vjp_team_kernel(const float* __restrict__ pack, int M, int S,
                                                            const float* __restrict__ x, const float* __restrict__ a, int R,
                                                            float* __restrict__ gx, int prior_only, Draws dw) {
  __shared__ float slots[2 * EV::kTeam * TeamCombine::DP];
  pack += blockIdx.y * dw.pack; x += blockIdx.y * dw.in; a += blockIdx.y * dw.in2; gx += blockIdx.y * dw.out;   // blockIdx.y = Monte-Carlo draw
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  EV ev;
  ev.init(pack, M, S, slots, wave, lane);
  for (int n = blockIdx.x; n < R; n += gridDim.x) {
    float xv[DI], av[DO], g[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) xv[i] = x[(size_t)n * DI + i];
#pragma unroll
    for (int i = 0; i < DO; ++i) av[i] = a[(size_t)n * DO + i];
    ev.vjp(xv, av, g, prior_only != 0);
    if (wave == 0) store_vec<DI>(gx + (size_t)n * DI, g, lane);
  }
}

// ---------------------------------------------------------------------------------------------
// B: parameter gradients in pack layout
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void st4(float* __restrict__ base, size_t f4_index, const float* v) {
  reinterpret_cast<float4*>(base)[f4_index] = make_float4(v[0], v[1], v[2], v[3]);
}

// this workgroup's gradient sums -> its slab (pack layout); shared by the parameter-sum kernel and the PGRAD reverse sweep
template <int DI, int DO, int NJ>
__device__ __forceinline__ void store_grads_rbf(const typename RbfTeamEval<DI, DO, NJ>::Grads& G, float* __restrict__ out, int M, int S,
                                                int wave, int lane, float (*sInd)[64][4 * RbfLayout<DI, DO>::RQ2],
                                                float (*sUni)[((DO + 1) / 2) * DI]) {
  using L = RbfLayout<DI, DO>;
  constexpr int DH = (DO + 1) / 2;
  const int SJ = cdiv(S, 64), MJ = cdiv(M, 64);
#pragma unroll
  for (int jn = 0; jn < NJ; ++jn) {
    const int j = wave + TEAM * jn;
    if (j < SJ) {
#pragma unroll
      for (int d = 0; d < DO; ++d)
#pragma unroll
        for (int q = 0; q < L::RQ; ++q) st4(out, (size_t)((j * DO + d) * L::RQ + q) * 64 + lane, &G.rff[jn * DO + d][4 * q]);
    }
  }
  // inducing record j = wave>>1: the two halves share the z fields -> add through LDS, lower half stores
  const int j = wave >> 1, half = wave & 1;
  if (half == 1) {
#pragma unroll
    for (int q = 0; q < 4 * L::RQ2; ++q) sInd[j][lane][q] = G.ind[q];
  }
  // uniform tail: lane-reduce this wave's partial of d/d wl[half*DH+dd][i]
  {
    float flat[DH * DI], red[DH * DI];
#pragma unroll
    for (int e = 0; e < DH * DI; ++e) flat[e] = G.gwl[e / DI][e % DI];
    wave_sum_all<DH * DI>(flat, red);
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < DH * DI; ++e) sUni[wave][e] = red[e];
    }
  }
  __syncthreads();
  if (half == 0 && j < MJ) {
    float* ind_out = out + 4 * L::rff_f4(S);
    float v[4 * L::RQ2];
#pragma unroll
    for (int q = 0; q < 4 * L::RQ2; ++q) v[q] = G.ind[q] + sInd[j][lane][q];
#pragma unroll
    for (int q = 0; q < L::RQ2; ++q) st4(ind_out, (size_t)(j * L::RQ2 + q) * 64 + lane, &v[4 * q]);
  }
  if (threadIdx.x < DO * DI) {
    const int d = threadIdx.x / DI, i = threadIdx.x % DI;
    const int h = d / DH, dd = d % DH;
    float* uni_out = out + 4 * (L::rff_f4(S) + L::ind_f4(M));
    uni_out[d * DI + i] = sUni[h][dd * DI + i] + sUni[h + 2][dd * DI + i];
  }
}

template <int DI, int DO, int NJ>
__global__ __launch_bounds__(256) void param_grad_rbf_kernel(const float* __restrict__ pack, int M, int S,
                                                              const float* __restrict__ xr, const float* __restrict__ ar,
                                                              int R, int rows_per_chunk, float* __restrict__ slab,
                                                              size_t pack_floats, int prior_only, Draws dw, int slab_chunks) {
  // blockIdx.y = Monte-Carlo draw: its pack, its rows, its chunk slabs
  pack += blockIdx.y * dw.pack; xr += blockIdx.y * dw.in; ar += blockIdx.y * dw.in2; slab += (size_t)blockIdx.y * slab_chunks * pack_floats;
  using EV = RbfTeamEval<DI, DO, NJ>;
  using L = RbfLayout<DI, DO>;
  constexpr int DH = (DO + 1) / 2;
  __shared__ float slots[2 * TEAM * TeamCombine::DP];
  __shared__ __attribute__((aligned(16))) float sInd[2][64][4 * L::RQ2];
  __shared__ float sUni[TEAM][DH * DI];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  EV ev;
  ev.init(pack, M, S, slots, wave, lane);
  typename EV::Grads G;
  G.zero();
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = min(R, r0 + rows_per_chunk);
  for (int r = r0; r < r1; ++r) {
    float x[DI], a[DO];
#pragma unroll
    for (int i = 0; i < DI; ++i) x[i] = xr[(size_t)r * DI + i];
#pragma unroll
    for (int i = 0; i < DO; ++i) a[i] = ar[(size_t)r * DO + i];
    ev.grad_row(x, a, G, prior_only != 0);
  }
  store_grads_rbf<DI, DO, NJ>(G, slab + (size_t)blockIdx.x * pack_floats, M, S, wave, lane, sInd, sUni);
}

template <int D, int NJ, int PART>
__device__ __forceinline__ void store_grads_df(const typename DfTeamEval<D, NJ>::Grads& G, float* __restrict__ out, int M, int S, int wave,
                                               int lane, float (*sInd)[64][4 * DfLayout<D>::RQ2],
                                               float (*sUni)[2 * D * ((D + 1) / 2) + (D + 1) / 2]) {
  using L = DfLayout<D>;
  constexpr int DH = (D + 1) / 2;
  constexpr int NU = 2 * D * DH + DH;
  const int SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  if constexpr (PART != 2) {
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
      const int j = wave + TEAM * jn;
      if (j < SJ) {
#pragma unroll
        for (int i = 0; i < D; ++i)
#pragma unroll
          for (int q = 0; q < L::RQ; ++q) st4(out, (size_t)((j * D + i) * L::RQ + q) * 64 + lane, &G.rff[jn * D + i][4 * q]);
      }
    }
  }
  if constexpr (PART == 1) return;                   // the inducing records and the uniform tail are another workgroup's
  const int j = wave >> 1, half = wave & 1;
  if (half == 1) {
#pragma unroll
    for (int q = 0; q < 4 * L::RQ2; ++q) sInd[j][lane][q] = G.ind[q];
  }
  {
    // lane-reduce the uniform partials in chunks of <= 8
    float flat[NU];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < DH; ++b) { flat[a * DH + b] = G.gwab[a][b]; flat[D * DH + a * DH + b] = G.gil2[a][b]; }
#pragma unroll
    for (int b = 0; b < DH; ++b) flat[2 * D * DH + b] = G.gvar[b];
    float red[NU];
    wave_sum_all<NU>(flat, red);
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < NU; ++e) sUni[wave][e] = red[e];
    }
  }
  __syncthreads();
  if (half == 0 && j < MJ) {
    float* ind_out = out + 4 * L::rff_f4(S);
    float v[4 * L::RQ2];
#pragma unroll
    for (int q = 0; q < 4 * L::RQ2; ++q) v[q] = G.ind[q] + sInd[j][lane][q];
#pragma unroll
    for (int q = 0; q < L::RQ2; ++q) st4(ind_out, (size_t)(j * L::RQ2 + q) * 64 + lane, &v[4 * q]);
  }
  {
    float* uni_out = out + 4 * (L::rff_f4(S) + L::ind_f4(M));
    const int t = threadIdx.x;
    if (t < D * D) {  // wab and il2: entry (a,b)
      const int a = t / D, b = t % D, h = b / DH, bb = b % DH;
      uni_out[a * D + b] = sUni[h][a * DH + bb] + sUni[h + 2][a * DH + bb];
      uni_out[D * D + a * D + b] = sUni[h][D * DH + a * DH + bb] + sUni[h + 2][D * DH + a * DH + bb];
    }
    if (t < D) {
      const int h = t / DH, bb = t % DH;
      uni_out[2 * D * D + t] = sUni[h][2 * D * DH + bb] + sUni[h + 2][2 * D * DH + bb];
    }
  }
}

template <int D, int NJ>
__global__ __launch_bounds__(256) void param_grad_df_kernel(const float* __restrict__ pack, int M, int S,
                                                             const float* __restrict__ xr, const float* __restrict__ ar,
                                                             int R, int rows_per_chunk, float* __restrict__ slab,
                                                             size_t pack_floats, int prior_only, Draws dw, int slab_chunks) {
  // blockIdx.y = Monte-Carlo draw: its pack, its rows, its chunk slabs
  pack += blockIdx.y * dw.pack; xr += blockIdx.y * dw.in; ar += blockIdx.y * dw.in2; slab += (size_t)blockIdx.y * slab_chunks * pack_floats;
  using EV = DfTeamEval<D, NJ>;
  using L = DfLayout<D>;
  constexpr int DH = (D + 1) / 2;
  constexpr int NU = 2 * D * DH + DH;  // gwab, gil2, gvar partials of one wave
  __shared__ float slots[2 * TEAM * TeamCombine::DP];
  __shared__ __attribute__((aligned(16))) float sInd[2][64][4 * L::RQ2];
  __shared__ float sUni[TEAM][NU];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  EV ev;
  ev.init(pack, M, S, slots, wave, lane);
  typename EV::Grads G;
  G.zero();
  const int r0 = blockIdx.x * rows_per_chunk;
  const int r1 = min(R, r0 + rows_per_chunk);
  for (int r = r0; r < r1; ++r) {
    float x[D], a[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { x[i] = xr[(size_t)r * D + i]; a[i] = ar[(size_t)r * D + i]; }
    ev.grad_row(x, a, G, prior_only != 0);
  }
  store_grads_df<D, NJ>(G, slab + (size_t)blockIdx.x * pack_floats, M, S, wave, lane, sInd, sUni);
}

// The same sums with the PACK split over two workgroups per chunk of rows (blockIdx.z: 0 = the Fourier-feature records, 1 = the
// inducing records and the uniform tail): either half keeps under 256 registers, so the two share a CU -- the unsplit kernel's 432
// registers allow one wavefront per SIMD, whose every dependent instruction waits out its full latency (168 us at configs[1], the
// first and longest node of the backward pass's side branch).
template <int D, int PART>
__device__ __forceinline__ void pgrad_df_part(const float* __restrict__ pack, int M, int S, const float* __restrict__ xr,
                                              const float* __restrict__ ar, int r0, int r1, float* __restrict__ out, int prior_only) {
  using EV = DfTeamEval<D, 1>;
  using L = DfLayout<D>;
  constexpr int DH = (D + 1) / 2;
  constexpr int NU = 2 * D * DH + DH;
  __shared__ float slots[2 * TEAM * TeamCombine::DP];
  __shared__ __attribute__((aligned(16))) float sInd[2][64][4 * L::RQ2];
  __shared__ float sUni[TEAM][NU];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  EV ev;
  ev.init(pack, M, S, slots, wave, lane);
  typename EV::Grads G;
  G.zero();
  for (int r = r0; r < r1; ++r) {
    float x[D], a[D];
#pragma unroll
    for (int i = 0; i < D; ++i) { x[i] = xr[(size_t)r * D + i]; a[i] = ar[(size_t)r * D + i]; }
    ev.template grad_row_part<PART>(x, a, G, prior_only != 0);
  }
  store_grads_df<D, 1, PART>(G, out, M, S, wave, lane, sInd, sUni);
}
template <int D>
__global__ __launch_bounds__(256, 2) void param_grad_df_split_kernel(const float* __restrict__ pack, int M, int S,
                                                                      const float* __restrict__ xr, const float* __restrict__ ar,
                                                                      int R, int rows_per_chunk, float* __restrict__ slab,
                                                                      size_t pack_floats, int prior_only, Draws dw, int slab_chunks) {
  pack += blockIdx.y * dw.pack; xr += blockIdx.y * dw.in; ar += blockIdx.y * dw.in2; slab += (size_t)blockIdx.y * slab_chunks * pack_floats;
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float* out = slab + (size_t)blockIdx.x * pack_floats;
  if (blockIdx.z == 0) pgrad_df_part<D, 1>(pack, M, S, xr, ar, r0, r1, out, prior_only);
  else pgrad_df_part<D, 2>(pack, M, S, xr, ar, r0, r1, out, prior_only);
}

// ---------------------------------------------------------------------------------------------
// B (streamed): the same sums for shapes past the register-resident team (S > 256, M > 128, DF D = 16).
// A workgroup still owns a chunk of rows, but walks the records one batch of four wavefronts at a time: a
// wavefront holds ONE record and that record's gradient in registers, streams the chunk's rows past it (x, a are
// wave-uniform loads from L2) and writes the finished record gradient to the slab; the uniform-tail partials stay
// in registers across the inducing batches.
// ---------------------------------------------------------------------------------------------
template <int DI, int DO>
__global__ __launch_bounds__(256) void param_grad_rbf_stream_kernel(const float* __restrict__ pack, int M, int S,
                                                                     const float* __restrict__ xr, const float* __restrict__ ar,
                                                                     int R, int rows_per_chunk, float* __restrict__ slab,
                                                                     size_t pack_floats, int prior_only, Draws dw, int slab_chunks) {
  // blockIdx.y = Monte-Carlo draw: its pack, its rows, its chunk slabs
  pack += blockIdx.y * dw.pack; xr += blockIdx.y * dw.in; ar += blockIdx.y * dw.in2; slab += (size_t)blockIdx.y * slab_chunks * pack_floats;
  using L = RbfLayout<DI, DO>;
  constexpr int DH = (DO + 1) / 2;
  __shared__ __attribute__((aligned(16))) float sInd[2][64][4 * L::RQ2];
  __shared__ float sUni[TEAM][DH * DI];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float4* p4 = reinterpret_cast<const float4*>(pack);
  const float4* i4 = p4 + L::rff_f4(S);
  const float* wl = pack + 4 * (L::rff_f4(S) + L::ind_f4(M));
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float* out = slab + (size_t)blockIdx.x * pack_floats;
  const int SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  for (int rec = wave; rec < SJ * DO; rec += TEAM) {
    float4 rq[L::RQ];
    load_record<L::RQ>(p4, rec, lane, rq);
    float g[4 * L::RQ];
#pragma unroll
    for (int q = 0; q < 4 * L::RQ; ++q) g[q] = 0.f;
    const int d = rec % DO;
    for (int r = r0; r < r1; ++r) {
      float x[DI], gx[DI];
#pragma unroll
      for (int i = 0; i < DI; ++i) { x[i] = xr[(size_t)r * DI + i]; gx[i] = 0.f; }
      rbf_rff_bwd<DI, DO, true>(rq, x, ar[(size_t)r * DO + d], gx, g);
    }
#pragma unroll
    for (int q = 0; q < L::RQ; ++q) st4(out, ((size_t)rec * L::RQ + q) * 64 + lane, &g[4 * q]);
  }
  const int half = wave & 1;
  float gwl[DH][DI];
#pragma unroll
  for (int d = 0; d < DH; ++d)
#pragma unroll
    for (int i = 0; i < DI; ++i) gwl[d][i] = 0.f;
  float* ind_out = out + 4 * L::rff_f4(S);
  for (int jb = 0; jb < MJ; jb += 2) {
    const int j = jb + (wave >> 1);
    const bool valid = j < MJ;
    float4 ind[L::RQ2];
    float gi[4 * L::RQ2];
#pragma unroll
    for (int q = 0; q < 4 * L::RQ2; ++q) gi[q] = 0.f;
    if (valid) {
      load_record<L::RQ2>(i4, j, lane, ind);
      if (!prior_only) {
        for (int r = r0; r < r1; ++r) {
          asm volatile("" ::: "memory");   // as in the DF kernel below
          float x[DI], a[DO], gx[DI];
#pragma unroll
          for (int i = 0; i < DI; ++i) { x[i] = xr[(size_t)r * DI + i]; gx[i] = 0.f; }
#pragma unroll
          for (int i = 0; i < DO; ++i) a[i] = ar[(size_t)r * DO + i];
          rbf_ind_half_bwd<DI, DO, true>(ind, x, wl, half, a, gx, gi, gwl);
        }
      }
    }
    if (half == 1) {
#pragma unroll
      for (int q = 0; q < 4 * L::RQ2; ++q) sInd[wave >> 1][lane][q] = gi[q];
    }
    __syncthreads();
    if (half == 0 && valid) {
      float v[4 * L::RQ2];
#pragma unroll
      for (int q = 0; q < 4 * L::RQ2; ++q) v[q] = gi[q] + sInd[wave >> 1][lane][q];
#pragma unroll
      for (int q = 0; q < L::RQ2; ++q) st4(ind_out, ((size_t)j * L::RQ2 + q) * 64 + lane, &v[4 * q]);
    }
    __syncthreads();
  }
  {
    float flat[DH * DI], red[DH * DI];
#pragma unroll
    for (int e = 0; e < DH * DI; ++e) flat[e] = gwl[e / DI][e % DI];
    wave_sum_all<DH * DI>(flat, red);
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < DH * DI; ++e) sUni[wave][e] = red[e];
    }
  }
  __syncthreads();
  if (threadIdx.x < DO * DI) {
    const int d = threadIdx.x / DI, i = threadIdx.x % DI;
    const int h = d / DH, dd = d % DH;
    float* uni_out = out + 4 * (L::rff_f4(S) + L::ind_f4(M));
    uni_out[d * DI + i] = sUni[h][dd * DI + i] + sUni[h + 2][dd * DI + i];
  }
}

// NP: column parts per inducing record (one wavefront each): 2 (halves) up to D = 8, 4 at D = 16 so that the uniform-parameter
// partials (2 x D x ceil(D/NP) registers) stay out of scratch
template <int D, int NP>
__global__ __launch_bounds__(256) void param_grad_df_stream_kernel(const float* __restrict__ pack, int M, int S,
                                                                    const float* __restrict__ xr, const float* __restrict__ ar,
                                                                    int R, int rows_per_chunk, float* __restrict__ slab,
                                                                    size_t pack_floats, int prior_only, Draws dw, int slab_chunks) {
  // blockIdx.y = Monte-Carlo draw: its pack, its rows, its chunk slabs
  pack += blockIdx.y * dw.pack; xr += blockIdx.y * dw.in; ar += blockIdx.y * dw.in2; slab += (size_t)blockIdx.y * slab_chunks * pack_floats;
  using L = DfLayout<D>;
  constexpr int DQ = (D + NP - 1) / NP;
  constexpr int NU = 2 * D * DQ + DQ;
  constexpr int RPI = TEAM / NP;                     // records per iteration of the inducing loop
  static_assert(TEAM % NP == 0, "column parts per team");
  __shared__ __attribute__((aligned(16))) float sInd[RPI][NP - 1][64][4 * L::RQ2];
  __shared__ float sUni[TEAM][NU];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const float4* p4 = reinterpret_cast<const float4*>(pack);
  const float4* i4 = p4 + L::rff_f4(S);
  const float* uni = pack + 4 * (L::rff_f4(S) + L::ind_f4(M));
  const int r0 = blockIdx.x * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float* out = slab + (size_t)blockIdx.x * pack_floats;
  const int SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  for (int rec = wave; rec < SJ * D; rec += TEAM) {
    float4 rq[L::RQ];
    load_record<L::RQ>(p4, rec, lane, rq);
    float g[4 * L::RQ];
#pragma unroll
    for (int q = 0; q < 4 * L::RQ; ++q) g[q] = 0.f;
    for (int r = r0; r < r1; ++r) {
      float x[D], a[D], gx[D];
#pragma unroll
      for (int i = 0; i < D; ++i) { x[i] = xr[(size_t)r * D + i]; a[i] = ar[(size_t)r * D + i]; gx[i] = 0.f; }
      df_rff_bwd<D, true>(rq, x, a, gx, g);
    }
#pragma unroll
    for (int q = 0; q < L::RQ; ++q) st4(out, ((size_t)rec * L::RQ + q) * 64 + lane, &g[4 * q]);
  }
  const int part = wave % NP, slot = wave / NP;
  float gwab[D][DQ], gil2[D][DQ], gvar[DQ];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int b = 0; b < DQ; ++b) { gwab[a][b] = 0.f; gil2[a][b] = 0.f; }
#pragma unroll
  for (int b = 0; b < DQ; ++b) gvar[b] = 0.f;
  float* ind_out = out + 4 * L::rff_f4(S);
  for (int jb = 0; jb < MJ; jb += RPI) {
    const int j = jb + slot;
    const bool valid = j < MJ;
    float4 ind[L::RQ2];
    float gi[4 * L::RQ2];
#pragma unroll
    for (int q = 0; q < 4 * L::RQ2; ++q) gi[q] = 0.f;
    if (valid) {
      load_record<L::RQ2>(i4, j, lane, ind);
      if (!prior_only) {
        for (int r = r0; r < r1; ++r) {
          asm volatile("" ::: "memory");   // uniform-table loads stay in the loop (hoisted, 2 D^2 + D values spill)
          float x[D], a[D], gx[D];
#pragma unroll
          for (int i = 0; i < D; ++i) { x[i] = xr[(size_t)r * D + i]; a[i] = ar[(size_t)r * D + i]; gx[i] = 0.f; }
          df_ind_part_bwd<D, NP, true>(ind, x, uni, part, a, gx, gi, gwab, gil2, gvar);
        }
      }
    }
    if (part > 0) {
#pragma unroll
      for (int q = 0; q < 4 * L::RQ2; ++q) sInd[slot][part - 1][lane][q] = gi[q];
    }
    __syncthreads();
    if (part == 0 && valid) {
      float v[4 * L::RQ2];
#pragma unroll
      for (int q = 0; q < 4 * L::RQ2; ++q) {
        v[q] = gi[q];
#pragma unroll
        for (int pp = 0; pp < NP - 1; ++pp) v[q] += sInd[slot][pp][lane][q];
      }
#pragma unroll
      for (int q = 0; q < L::RQ2; ++q) st4(ind_out, ((size_t)j * L::RQ2 + q) * 64 + lane, &v[4 * q]);
    }
    __syncthreads();
  }
  {
    float flat[NU];
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int b = 0; b < DQ; ++b) { flat[a * DQ + b] = gwab[a][b]; flat[D * DQ + a * DQ + b] = gil2[a][b]; }
#pragma unroll
    for (int b = 0; b < DQ; ++b) flat[2 * D * DQ + b] = gvar[b];
    float red[NU];
    wave_sum_all<NU>(flat, red);
    if (lane == 0) {
#pragma unroll
      for (int e = 0; e < NU; ++e) sUni[wave][e] = red[e];
    }
  }
  __syncthreads();
  {
    // column b belongs to part b / DQ; the wavefronts w = part, part + NP, ... hold its partials
    float* uni_out = out + 4 * (L::rff_f4(S) + L::ind_f4(M));
    for (int t = threadIdx.x; t < D * D; t += 256) {
      const int a = t / D, b = t % D, pt = b / DQ, bb = b % DQ;
      float v0 = 0.f, v1 = 0.f;
      for (int w = pt; w < TEAM; w += NP) { v0 += sUni[w][a * DQ + bb]; v1 += sUni[w][D * DQ + a * DQ + bb]; }
      uni_out[a * D + b] = v0;
      uni_out[D * D + a * D + b] = v1;
    }
    if (threadIdx.x < D) {
      const int t = threadIdx.x, pt = t / DQ, bb = t % DQ;
      float v = 0.f;
      for (int w = pt; w < TEAM; w += NP) v += sUni[w][2 * D * DQ + bb];
      uni_out[2 * D * D + t] = v;
    }
  }
}

// gpack[e] = sum_c slab[c][e] in a fixed order: 16 interleaved partial sums per element (independent load streams),
// combined through LDS.  grid = ceil(pack_floats / 64), block = 1024 (64 elements x 16 chunk groups).
__global__ __launch_bounds__(1024) void reduce_slab_kernel(const float* __restrict__ slab, int nchunk, size_t pack_floats,
                                                            float* __restrict__ gpack, int accumulate, size_t gpack_dstride, int slab_chunks) {
  __shared__ float red[16][64];
  slab += (size_t)blockIdx.y * slab_chunks * pack_floats; gpack += blockIdx.y * gpack_dstride;   // blockIdx.y = Monte-Carlo draw
  const int ex = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const size_t e = (size_t)blockIdx.x * 64 + ex;
  float acc = 0.f;
  if (e < pack_floats) {
#pragma unroll 8
    for (int c = cg; c < nchunk; c += 16) acc += slab[(size_t)c * pack_floats + e];
  }
  red[cg][ex] = acc;
  __syncthreads();
  if (cg == 0 && e < pack_floats) {
    float v = accumulate ? gpack[e] : 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += red[g][ex];
    gpack[e] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline int team_grid_b(int N) { return N < 2048 ? N : 2048; }
// wavefronts per trajectory of the wide team (gp_wide.hpp) for an output width, 0: none
template <int DO> constexpr int wide_ts_b() { return (DO == 6 || DO == 3) ? 12 : (DO == 4 ? 8 : (DO == 8 ? 16 : 0)); }

// register-resident team when the quarter pack fits (S <= 256, M <= 128, D <= 8), streamed team otherwise
template <int DI, int DO> static bool rbf_team_ok(int M, int S) {
  if constexpr (DO <= 8) return RbfTeamEval<DI, DO, 1>::fits(M, S);
  return false;
}
template <int D> static bool df_team_ok(int M, int S) {
  if constexpr (D <= 8) return DfTeamEval<D, 1>::fits(M, S);
  return false;
}

template <int DI, int DO, int ORDER, int METHOD>
static int launch_bwd_rbf(const float* pack, int M, int S, const float* xstage, const float* gzt, const float* ts, int N, int T,
                          float* gz0, float* astage, hipStream_t st, Draws dw) {
  if constexpr (wide_ts_b<DO>() > 0 && DI <= 8) {   // few trajectories: 12 wavefronts each (gp_wide.hpp)
    constexpr int TS = wide_ts_b<DO>();
    if (wide_team_enabled() && N <= kWideMaxRows && RbfWideTeam<DI, DO, TS>::fits(M, S)) {
      hipLaunchKernelGGL((rollout_bwd_team_kernel<RbfWideTeam<DI, DO, TS>, DI, DO, ORDER, METHOD>), dim3(N, dw.nd), 64 * TS, 0, st,
                         pack, M, S, xstage, gzt, ts, N, T, gz0, astage, dw);
      return check_launch("rollout_bwd_rbf_wide");
    }
  }
  if constexpr (DO <= 8) {
    if (rbf_team_ok<DI, DO>(M, S)) {
      hipLaunchKernelGGL((rollout_bwd_team_kernel<RbfTeamEval<DI, DO, 1>, DI, DO, ORDER, METHOD>), dim3(team_grid_b(N), dw.nd), 256, 0, st,
                         pack, M, S, xstage, gzt, ts, N, T, gz0, astage, dw);
      return check_launch("rollout_bwd_rbf");
    }
  }
  hipLaunchKernelGGL((rollout_bwd_team_kernel<RbfStreamTeam<DI, DO>, DI, DO, ORDER, METHOD>), dim3(team_grid_b(N), dw.nd), 256, 0, st,
                     pack, M, S, xstage, gzt, ts, N, T, gz0, astage, dw);
  return check_launch("rollout_bwd_rbf_stream");
}

template <int D, int METHOD>
static int launch_bwd_df(const float* pack, int M, int S, const float* xstage, const float* gzt, const float* ts, int N, int T,
                         float* gz0, float* astage, hipStream_t st, Draws dw) {
  if constexpr (wide_ts_b<D>() > 0) {
    constexpr int TS = wide_ts_b<D>();
    if (wide_team_enabled() && N <= kWideMaxRows && DfWideTeam<D, TS>::fits(M, S)) {
      hipLaunchKernelGGL((rollout_bwd_team_kernel<DfWideTeam<D, TS>, D, D, 1, METHOD>), dim3(N, dw.nd), 64 * TS, 0, st,
                         pack, M, S, xstage, gzt, ts, N, T, gz0, astage, dw);
      return check_launch("rollout_bwd_df_wide");
    }
  }
  if constexpr (D <= 8) {
    if (df_team_ok<D>(M, S)) {
      hipLaunchKernelGGL((rollout_bwd_team_kernel<DfTeamEval<D, 1>, D, D, 1, METHOD>), dim3(team_grid_b(N), dw.nd), 256, 0, st,
                         pack, M, S, xstage, gzt, ts, N, T, gz0, astage, dw);
      return check_launch("rollout_bwd_df");
    }
  }
  hipLaunchKernelGGL((rollout_bwd_team_kernel<DfStreamTeam<D>, D, D, 1, METHOD>), dim3(team_grid_b(N), dw.nd), 256, 0, st,
                     pack, M, S, xstage, gzt, ts, N, T, gz0, astage, dw);
  return check_launch("rollout_bwd_df_stream");
}

#define GP_BWD_RBF_DIMS(X) X(6, 6) X(6, 3) X(4, 4) X(4, 2) X(2, 2) X(2, 1) X(8, 8) X(8, 4) X(3, 3) X(16, 16) X(16, 8) X(12, 6)
#define GP_BWD_DF_DIMS(X) X(6) X(4) X(2) X(3) X(8) X(16) X(5) X(7) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int DI, int DO>
static int bwd_rbf_dispatch(int order, int method, const float* pack, int M, int S, const float* xstage, const float* gzt,
                            const float* ts, int N, int T, float* gz0, float* astage, hipStream_t st, Draws dw) {
  if constexpr (DI == DO) {
    if (order == 1 && method == 0) return launch_bwd_rbf<DI, DO, 1, 0>(pack, M, S, xstage, gzt, ts, N, T, gz0, astage, st, dw);
    if (order == 1 && method == 1) return launch_bwd_rbf<DI, DO, 1, 1>(pack, M, S, xstage, gzt, ts, N, T, gz0, astage, st, dw);
    if (order == 1 && method == 2) return launch_bwd_rbf<DI, DO, 1, 2>(pack, M, S, xstage, gzt, ts, N, T, gz0, astage, st, dw);
  }
  if constexpr (DI == 2 * DO) {
    if (order == 2 && method == 0) return launch_bwd_rbf<DI, DO, 2, 0>(pack, M, S, xstage, gzt, ts, N, T, gz0, astage, st, dw);
    if (order == 2 && method == 1) return launch_bwd_rbf<DI, DO, 2, 1>(pack, M, S, xstage, gzt, ts, N, T, gz0, astage, st, dw);
    if (order == 2 && method == 2) return launch_bwd_rbf<DI, DO, 2, 2>(pack, M, S, xstage, gzt, ts, N, T, gz0, astage, st, dw);
  }
  return set_error("gpode_rollout_bwd: order=%d needs Di == order*Do (Di=%d Do=%d)", order, DI, DO);
}

int rollout_bwd(int kernel, int order, int method, int Di, int Do, int M, int S, const float* pack, const float* xstage,
                const float* gzt, const float* ts, int N, int T, float* gz0, float* astage, hipStream_t st, Draws dw) {
  if (method < 0 || method > 2) return set_error("gpode_rollout_bwd: method %d", method);
  if (kernel == 0) {
#define X(a, b) if (Di == a && Do == b) return bwd_rbf_dispatch<a, b>(order, method, pack, M, S, xstage, gzt, ts, N, T, gz0, astage, st, dw);
    GP_BWD_RBF_DIMS(X)
#undef X
  } else {
    if (order != 1) return set_error("gpode_rollout_bwd: DF kernel is first-order only");
#define X(a) if (Di == a && Do == a) return method == 0 ? launch_bwd_df<a, 0>(pack, M, S, xstage, gzt, ts, N, T, gz0, astage, st, dw) \
                                            : method == 1 ? launch_bwd_df<a, 1>(pack, M, S, xstage, gzt, ts, N, T, gz0, astage, st, dw) \
                                                          : launch_bwd_df<a, 2>(pack, M, S, xstage, gzt, ts, N, T, gz0, astage, st, dw);
    GP_BWD_DF_DIMS(X)
#undef X
  }
  return set_error("gpode_rollout_bwd: no specialisation for kernel=%d Di=%d Do=%d", kernel, Di, Do);
}

// ---- reverse sweep + parameter sums in one pass (rollout_bwd_team_kernel<..., PG = true>) --------------------------------------
// Built for the first-order register-resident teams of the widths below and the Euler / 3-8-rule solvers; pgrad_chunks() = 0 for
// every other shape: the caller then runs rollout_bwd and param_grad separately.
// (RBF widths up to 6: 376 registers at q = 6 with the 3/8 rule.  q = 8 spills 181 registers, and the divergence-free kernel, whose
//  parameter-sum kernel alone takes 432, spills 945 -- besides, at configs[1] its sums (168 us) sit on the side branch, and moving them
//  into the main branch's sweep lengthens the chain the encoder's backward follows by about what the side branch gets shorter)
#define GP_PGRAD_RBF_DIMS(X) X(6) X(4) X(2) X(3)
int rollout_bwd_pgrad_chunks(int kernel, int order, int method, int Di, int Do, int M, int S, int N) {
  if (order != 1 || Di != Do || (method != 0 && method != 1) || N < 1) return 0;
  if (kernel == 0) {
#define X(d) if (Do == d) return rbf_team_ok<d, d>(M, S) ? team_grid_b(N) : 0;
    GP_PGRAD_RBF_DIMS(X)
#undef X
  }
  return 0;
}
int rollout_bwd_pgrad(int kernel, int order, int method, int Di, int Do, int M, int S, const float* pack, const float* xstage,
                      const float* gzt, const float* ts, int N, int T, float* gz0, float* astage, float* slab, int nchunk,
                      float* gpack, hipStream_t st, Draws dw) {
  const int grid = rollout_bwd_pgrad_chunks(kernel, order, method, Di, Do, M, S, N);
  if (grid == 0) return set_error("gpode_rollout_bwd_pgrad: no fused form for kernel=%d order=%d method=%d Di=%d Do=%d M=%d S=%d", kernel, order,
                                  method, Di, Do, M, S);
  if (nchunk != grid) return set_error("gpode_rollout_bwd_pgrad: slab for %d chunks, the launch has %d", nchunk, grid);
  size_t pf = 0;
  if (cache_sizes(kernel, Di, Do, M, S, &pf, nullptr)) return 1;
  bool done = false;
  if (kernel == 0) {
#define X(d)                                                                                                                         \
  if (Do == d) {                                                                                                                     \
    if (method == 0) hipLaunchKernelGGL((rollout_bwd_team_kernel<RbfTeamEval<d, d, 1>, d, d, 1, 0, true>), dim3(grid, dw.nd), 256, 0, st, \
                                        pack, M, S, xstage, gzt, ts, N, T, gz0, astage, dw, slab, pf);                               \
    else hipLaunchKernelGGL((rollout_bwd_team_kernel<RbfTeamEval<d, d, 1>, d, d, 1, 1, true>), dim3(grid, dw.nd), 256, 0, st,        \
                            pack, M, S, xstage, gzt, ts, N, T, gz0, astage, dw, slab, pf);                                           \
    done = true;                                                                                                                     \
  }
    GP_PGRAD_RBF_DIMS(X)
#undef X
  }
  if (!done) return set_error("gpode_rollout_bwd_pgrad: width %d not built", Do);
  if (check_launch("rollout_bwd_pgrad")) return 1;
  // the chunk slabs -> the pack-layout gradient of every draw (dw.out2 is astage's draw stride; gpack is dense per draw)
  hipLaunchKernelGGL(reduce_slab_kernel, dim3((unsigned)((pf + 63) / 64), dw.nd), 1024, 0, st, slab, grid, pf, gpack, 0, pf, grid);
  return check_launch("reduce_slab");
}

template <int DI, int DO>
static int launch_vjp_rbf(const float* pack, int M, int S, const float* x, const float* a, int R, float* gx, int prior_only, hipStream_t st, Draws dw) {
  if constexpr (DO <= 8) {
    if (rbf_team_ok<DI, DO>(M, S)) {
      hipLaunchKernelGGL((rhs_vjp_team_kernel<RbfTeamEval<DI, DO, 1>, DI, DO>), dim3(team_grid_b(R), dw.nd), 256, 0, st, pack, M, S, x, a, R, gx, prior_only, dw);
      return check_launch("rhs_vjp_rbf");
    }
  }
  hipLaunchKernelGGL((rhs_vjp_team_kernel<RbfStreamTeam<DI, DO>, DI, DO>), dim3(team_grid_b(R), dw.nd), 256, 0, st, pack, M, S, x, a, R, gx, prior_only, dw);
  return check_launch("rhs_vjp_rbf_stream");
}

template <int D>
static int launch_vjp_df(const float* pack, int M, int S, const float* x, const float* a, int R, float* gx, int prior_only, hipStream_t st, Draws dw) {
  if constexpr (D <= 8) {
    if (df_team_ok<D>(M, S)) {
      hipLaunchKernelGGL((rhs_vjp_team_kernel<DfTeamEval<D, 1>, D, D>), dim3(team_grid_b(R), dw.nd), 256, 0, st, pack, M, S, x, a, R, gx, prior_only, dw);
      return check_launch("rhs_vjp_df");
    }
  }
  hipLaunchKernelGGL((rhs_vjp_team_kernel<DfStreamTeam<D>, D, D>), dim3(team_grid_b(R), dw.nd), 256, 0, st, pack, M, S, x, a, R, gx, prior_only, dw);
  return check_launch("rhs_vjp_df_stream");
}

int rhs_vjp(int kernel, int Di, int Do, int M, int S, const float* pack, const float* x, const float* a, int R, float* gx,
            int prior_only, hipStream_t st, Draws dw) {
  if (R <= 0) return 0;
  if (kernel == 0) {
#define X(p, q) if (Di == p && Do == q) return launch_vjp_rbf<p, q>(pack, M, S, x, a, R, gx, prior_only, st, dw);
    GP_BWD_RBF_DIMS(X)
#undef X
  } else {
#define X(p) if (Di == p && Do == p) return launch_vjp_df<p>(pack, M, S, x, a, R, gx, prior_only, st, dw);
    GP_BWD_DF_DIMS(X)
#undef X
  }
  return set_error("gpode_rhs_vjp: no specialisation for kernel=%d Di=%d Do=%d", kernel, Di, Do);
}

// rows (R,Di) x adjoints (R,Do) -> gpack (pack layout).  slab: nchunk * pack_floats floats of scratch.
template <int DI, int DO>
static int launch_pgrad_rbf(const float* pack, int M, int S, const float* xr, const float* ar, int R, int rpc, int used, float* slab,
                            size_t pf, int prior_only, hipStream_t st, Draws dw, int nchunk) {
  if constexpr (DO <= 8) {
    if (rbf_team_ok<DI, DO>(M, S)) {
      hipLaunchKernelGGL((param_grad_rbf_kernel<DI, DO, 1>), dim3(used, dw.nd), 256, 0, st, pack, M, S, xr, ar, R, rpc, slab, pf, prior_only, dw, nchunk);
      return check_launch("param_grad_rbf");
    }
  }
  hipLaunchKernelGGL((param_grad_rbf_stream_kernel<DI, DO>), dim3(used, dw.nd), 256, 0, st, pack, M, S, xr, ar, R, rpc, slab, pf, prior_only, dw, nchunk);
  return check_launch("param_grad_rbf_stream");
}

template <int D>
static int launch_pgrad_df(const float* pack, int M, int S, const float* xr, const float* ar, int R, int rpc, int used, float* slab,
                           size_t pf, int prior_only, hipStream_t st, Draws dw, int nchunk) {
  if constexpr (D <= 8) {
    if (df_team_ok<D>(M, S)) {
      // many rows (the rows of a training step): the pack split over two co-resident workgroups per chunk; GPODE_PGRAD_DF_UNSPLIT=1: A/B
      static const bool unsplit = [] { const char* e = getenv("GPODE_PGRAD_DF_UNSPLIT"); return e && e[0] == '1'; }();
      if constexpr (D == 6) {
        if (!unsplit && R >= 1024) {
          hipLaunchKernelGGL((param_grad_df_split_kernel<D>), dim3(used, dw.nd, 2), 256, 0, st, pack, M, S, xr, ar, R, rpc, slab, pf, prior_only, dw, nchunk);
          return check_launch("param_grad_df_split");
        }
      }
      hipLaunchKernelGGL((param_grad_df_kernel<D, 1>), dim3(used, dw.nd), 256, 0, st, pack, M, S, xr, ar, R, rpc, slab, pf, prior_only, dw, nchunk);
      return check_launch("param_grad_df");
    }
  }
  hipLaunchKernelGGL((param_grad_df_stream_kernel<D, (D > 8 ? 4 : 2)>), dim3(used, dw.nd), 256, 0, st, pack, M, S, xr, ar, R, rpc, slab, pf, prior_only, dw, nchunk);
  return check_launch("param_grad_df_stream");
}

int param_grad(int kernel, int Di, int Do, int M, int S, const float* pack, const float* xr, const float* ar, int R,
               float* slab, int nchunk, float* gpack, int accumulate, int prior_only, hipStream_t st, Draws dw) {
  size_t pf = 0;
  if (cache_sizes(kernel, Di, Do, M, S, &pf, nullptr)) return 1;
  if (R <= 0 || nchunk <= 0) return set_error("gpode_param_grad: R=%d nchunk=%d", R, nchunk);
  const int rpc = cdiv(R, nchunk);
  const int used = cdiv(R, rpc);
  int rc = -1;
  if (kernel == 0) {
#define X(p, q) if (Di == p && Do == q) rc = launch_pgrad_rbf<p, q>(pack, M, S, xr, ar, R, rpc, used, slab, pf, prior_only, st, dw, nchunk);
    GP_BWD_RBF_DIMS(X)
#undef X
  } else {
#define X(p) if (Di == p && Do == p) rc = launch_pgrad_df<p>(pack, M, S, xr, ar, R, rpc, used, slab, pf, prior_only, st, dw, nchunk);
    GP_BWD_DF_DIMS(X)
#undef X
  }
  if (rc < 0) return set_error("gpode_param_grad: no specialisation for kernel=%d Di=%d Do=%d", kernel, Di, Do);
  if (rc) return rc;
  hipLaunchKernelGGL(reduce_slab_kernel, dim3((unsigned)((pf + 63) / 64), dw.nd), 1024, 0, st, slab, used, pf, gpack, accumulate, dw.out, nchunk);
  return check_launch("reduce_slab");
}

}  // namespace gp

// gp_team.hpp -- 4-wave TEAM evaluators shared by the forward (gp_forward.hip) and backward
// (gp_backward.hip) kernels: value f(x), vector-Jacobian product J(x)^T a, and per-row parameter
// gradients, all on the same register-resident quarter of the pack.
#pragma once
#include "gp_eval.hpp"

namespace gp {

// ----------------------------------------------------------------------------------------------
// 4-wave TEAM mapping (few trajectories: batch <~ 2048).
// One 256-thread workgroup = one trajectory at a time.  Wave w owns RFF lane-groups j = w, w+4, ...
// and the inducing work unit u = w (u = 2 j + half: record j, output-dim half), so its parameter slice
// is a quarter of the pack (cfg1: 60 floats per lane, cfg2: 108) and lives in VGPRs for the whole
// launch -- no spills, no LDS/L2 re-reads.  Per evaluation each wave reduces its partial f over its 64
// lanes (transposing reduction -> wave-uniform), lane 0 drops it in an LDS slot, ONE s_barrier, and all
// four waves sum the four slots in fixed order (deterministic).  Slots are double-buffered by
// evaluation parity: a wave can run at most one evaluation ahead of the slowest one.
// ----------------------------------------------------------------------------------------------
constexpr int TEAM = 4;

template <int TS> struct TeamCombineT {
  static constexpr int DP = 16;  // floats per wave slot (D <= 16 for every compiled specialisation)
  float* slots;                  // [2][TS][DP] in LDS
  int wave, lane, parity;
  __device__ __forceinline__ void init(float* s, int w, int l) { slots = s; wave = w; lane = l; parity = 0; }
  template <int NV> __device__ __forceinline__ void run(const float (&part)[NV], float (&f)[NV]) {
    static_assert(NV <= DP, "slot too small");
    float* mine = slots + (parity * TS + wave) * DP;
    if (lane == 0) {
#pragma unroll
      for (int d = 0; d < NV; ++d) mine[d] = part[d];
    }
    __syncthreads();
    const float* base = slots + parity * TS * DP;
#pragma unroll
    for (int d = 0; d < NV; ++d) {
      float v = base[d];
#pragma unroll
      for (int w = 1; w < TS; ++w) v += base[w * DP + d];
      f[d] = v;
    }
    parity ^= 1;
  }
};
using TeamCombine = TeamCombineT<TEAM>;

template <int DI, int DO, int NJ> struct RbfTeamEval {
  using L = RbfLayout<DI, DO>;
  static constexpr int kTeam = TEAM;
  float4 rff[NJ * DO][L::RQ];
  float4 ind[L::RQ2];
  const float* wl;
  int half;
  TeamCombine comb;
  static __host__ bool fits(int M, int S) { return cdiv(S, 64) <= TEAM * NJ && cdiv(M, 64) * 2 <= TEAM; }
  __device__ __forceinline__ void init(const float* pack, int M, int S, float* lds, int wave, int lane) {
    const float4* p4 = reinterpret_cast<const float4*>(pack);
    const int SJ = cdiv(S, 64), MJ = cdiv(M, 64);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
      const int j = wave + TEAM * jn;
#pragma unroll
      for (int d = 0; d < DO; ++d)
#pragma unroll
        for (int q = 0; q < L::RQ; ++q) rff[jn * DO + d][q] = j < SJ ? p4[((j * DO + d) * L::RQ + q) * 64 + lane] : z;
    }
    const float4* i4 = p4 + L::rff_f4(S);
    const int j = wave >> 1;
    half = wave & 1;
#pragma unroll
    for (int q = 0; q < L::RQ2; ++q) ind[q] = j < MJ ? i4[(j * L::RQ2 + q) * 64 + lane] : z;
    wl = pack + 4 * (L::rff_f4(S) + L::ind_f4(M));
    comb.init(lds, wave, lane);
  }
  template <int MODE> __device__ __forceinline__ void eval(const float (&x)[DI], float (&f)[DO]) {
    float acc[DO];
#pragma unroll
    for (int d = 0; d < DO; ++d) acc[d] = 0.f;
    if (MODE != 2) {
#pragma unroll
      for (int r = 0; r < NJ * DO; ++r) rbf_rff_record<DI, DO>(rff[r], x, acc[r % DO]);
    }
    if (MODE != 1) rbf_ind_record_half<DI, DO>(ind, x, wl, half, acc);
    float part[DO];
    wave_sum_all<DO>(acc, part);
    comb.run<DO>(part, f);
  }
  // gx = J_f(x)^T a, combined over the team
  // prior_only: differentiate f_prior alone (the f_prior(Z) path of the cache build)
  __device__ __forceinline__ void vjp(const float (&x)[DI], const float (&a)[DO], float (&gx)[DI], bool prior_only = false) {
    float acc[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) acc[i] = 0.f;
    float g0[4 * L::RQ], g1[4 * L::RQ2], g2[(DO + 1) / 2][DI];
#pragma unroll
    for (int r = 0; r < NJ * DO; ++r) rbf_rff_bwd<DI, DO, false>(rff[r], x, a[r % DO], acc, g0);
    if (!prior_only) rbf_ind_half_bwd<DI, DO, false>(ind, x, wl, half, a, acc, g1, g2);
    float part[DI];
    wave_sum_all<DI>(acc, part);
    comb.run<DI>(part, gx);
  }
  // per-row parameter gradients of this wave's slice (no cross-wave traffic)
  struct Grads {
    float rff[NJ * DO][4 * L::RQ];
    float ind[4 * L::RQ2];
    float gwl[(DO + 1) / 2][DI];
    __device__ __forceinline__ void zero() {
#pragma unroll
      for (int r = 0; r < NJ * DO; ++r)
#pragma unroll
        for (int q = 0; q < 4 * L::RQ; ++q) rff[r][q] = 0.f;
#pragma unroll
      for (int q = 0; q < 4 * L::RQ2; ++q) ind[q] = 0.f;
#pragma unroll
      for (int d = 0; d < (DO + 1) / 2; ++d)
#pragma unroll
        for (int i = 0; i < DI; ++i) gwl[d][i] = 0.f;
    }
  };
  __device__ __forceinline__ void grad_row(const float (&x)[DI], const float (&a)[DO], Grads& G, bool prior_only) const {
    float gx[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) gx[i] = 0.f;
#pragma unroll
    for (int r = 0; r < NJ * DO; ++r) rbf_rff_bwd<DI, DO, true>(rff[r], x, a[r % DO], gx, G.rff[r]);
    if (!prior_only) rbf_ind_half_bwd<DI, DO, true>(ind, x, wl, half, a, gx, G.ind, G.gwl);
  }
  // vjp() and grad_row() in one pass over the records: the reverse sweep of the integrator visits every (x, a) row anyway, and
  // the parameter sums need nothing else (one launch, and one pass over the rows, less)
  __device__ __forceinline__ void vjp_grad(const float (&x)[DI], const float (&a)[DO], float (&gx)[DI], Grads& G) {
    float acc[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) acc[i] = 0.f;
#pragma unroll
    for (int r = 0; r < NJ * DO; ++r) rbf_rff_bwd<DI, DO, true>(rff[r], x, a[r % DO], acc, G.rff[r]);
    rbf_ind_half_bwd<DI, DO, true>(ind, x, wl, half, a, acc, G.ind, G.gwl);
    float part[DI];
    wave_sum_all<DI>(acc, part);
    comb.run<DI>(part, gx);
  }
};

template <int D, int NJ> struct DfTeamEval {
  using L = DfLayout<D>;
  static constexpr int kTeam = TEAM;
  float4 rff[NJ * D][L::RQ];
  float4 ind[L::RQ2];
  const float* uni;
  int half;
  TeamCombine comb;
  static __host__ bool fits(int M, int S) { return cdiv(S, 64) <= TEAM * NJ && cdiv(M, 64) * 2 <= TEAM; }
  __device__ __forceinline__ void init(const float* pack, int M, int S, float* lds, int wave, int lane) {
    const float4* p4 = reinterpret_cast<const float4*>(pack);
    const int SJ = cdiv(S, 64), MJ = cdiv(M, 64);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int jn = 0; jn < NJ; ++jn) {
      const int j = wave + TEAM * jn;
#pragma unroll
      for (int i = 0; i < D; ++i)
#pragma unroll
        for (int q = 0; q < L::RQ; ++q) rff[jn * D + i][q] = j < SJ ? p4[((j * D + i) * L::RQ + q) * 64 + lane] : z;
    }
    const float4* i4 = p4 + L::rff_f4(S);
    const int j = wave >> 1;
    half = wave & 1;
#pragma unroll
    for (int q = 0; q < L::RQ2; ++q) ind[q] = j < MJ ? i4[(j * L::RQ2 + q) * 64 + lane] : z;
    uni = pack + 4 * (L::rff_f4(S) + L::ind_f4(M));
    comb.init(lds, wave, lane);
  }
  template <int MODE> __device__ __forceinline__ void eval(const float (&x)[D], float (&f)[D]) {
    float acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    if (MODE != 2) {
#pragma unroll
      for (int r = 0; r < NJ * D; ++r) df_rff_record<D>(rff[r], x, acc);
    }
    if (MODE != 1) df_ind_record_half<D>(ind, x, uni, half, acc);
    float part[D];
    wave_sum_all<D>(acc, part);
    comb.run<D>(part, f);
  }
  __device__ __forceinline__ void vjp(const float (&x)[D], const float (&a)[D], float (&gx)[D], bool prior_only = false) {
    float acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = 0.f;
    float g0[4 * L::RQ], g1[4 * L::RQ2], g2[D][(D + 1) / 2], g3[D][(D + 1) / 2], g4[(D + 1) / 2];
#pragma unroll
    for (int r = 0; r < NJ * D; ++r) df_rff_bwd<D, false>(rff[r], x, a, acc, g0);
    if (!prior_only) df_ind_half_bwd<D, false>(ind, x, uni, half, a, acc, g1, g2, g3, g4);
    float part[D];
    wave_sum_all<D>(acc, part);
    comb.run<D>(part, gx);
  }
  struct Grads {
    float rff[NJ * D][4 * L::RQ];
    float ind[4 * L::RQ2];
    float gwab[D][(D + 1) / 2], gil2[D][(D + 1) / 2], gvar[(D + 1) / 2];
    __device__ __forceinline__ void zero() {
#pragma unroll
      for (int r = 0; r < NJ * D; ++r)
#pragma unroll
        for (int q = 0; q < 4 * L::RQ; ++q) rff[r][q] = 0.f;
#pragma unroll
      for (int q = 0; q < 4 * L::RQ2; ++q) ind[q] = 0.f;
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int b = 0; b < (D + 1) / 2; ++b) { gwab[a][b] = 0.f; gil2[a][b] = 0.f; }
#pragma unroll
      for (int b = 0; b < (D + 1) / 2; ++b) gvar[b] = 0.f;
    }
  };
  __device__ __forceinline__ void grad_row(const float (&x)[D], const float (&a)[D], Grads& G, bool prior_only) const {
    float gx[D];
#pragma unroll
    for (int i = 0; i < D; ++i) gx[i] = 0.f;
#pragma unroll
    for (int r = 0; r < NJ * D; ++r) df_rff_bwd<D, true>(rff[r], x, a, gx, G.rff[r]);
    if (!prior_only) df_ind_half_bwd<D, true>(ind, x, uni, half, a, gx, G.ind, G.gwab, G.gil2, G.gvar);
  }
  // PART 1: the Fourier-feature records only; PART 2: the inducing records (and the uniform tail) only -- the two halves of
  // grad_row for workgroups that split the pack instead of the rows (param_grad_df_split_kernel)
  template <int PART> __device__ __forceinline__ void grad_row_part(const float (&x)[D], const float (&a)[D], Grads& G, bool prior_only) const {
    float gx[D];
#pragma unroll
    for (int i = 0; i < D; ++i) gx[i] = 0.f;
    if constexpr (PART != 2) {
#pragma unroll
      for (int r = 0; r < NJ * D; ++r) df_rff_bwd<D, true>(rff[r], x, a, gx, G.rff[r]);
    }
    if constexpr (PART != 1) {
      if (!prior_only) df_ind_half_bwd<D, true>(ind, x, uni, half, a, gx, G.ind, G.gwab, G.gil2, G.gvar);
    }
  }
  __device__ __forceinline__ void vjp_grad(const float (&x)[D], const float (&a)[D], float (&gx)[D], Grads& G) {
    float acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = 0.f;
#pragma unroll
    for (int r = 0; r < NJ * D; ++r) df_rff_bwd<D, true>(rff[r], x, a, acc, G.rff[r]);
    df_ind_half_bwd<D, true>(ind, x, uni, half, a, acc, G.ind, G.gwab, G.gil2, G.gvar);
    float part[D];
    wave_sum_all<D>(acc, part);
    comb.run<D>(part, gx);
  }
};

// ----------------------------------------------------------------------------------------------
// STREAMED team evaluators: the same team and combine (TS wavefronts, 4 in every launch), but the pack stays in
// global memory -- it is L2-resident (0.66 MB at D=16, M=512, S=256) -- and every wave walks its share
// of the records: rff records rec = wave, wave+4, ...; inducing work units u = 2 j + half = wave,
// wave+4, ... (so `half` is still a per-wave constant).  Any S and M, and the D whose records no longer
// fit a register-resident quarter (DF D=16: 36 floats per record, 16 records per lane group).
// ----------------------------------------------------------------------------------------------
template <int NQ>
__device__ __forceinline__ void load_record(const float4* __restrict__ base, int rec, int lane, float4 (&r)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) r[q] = base[((size_t)rec * NQ + q) * 64 + lane];
}

// v[idx] for a wave-uniform idx without a runtime register index
template <int N> __device__ __forceinline__ float pick(const float (&v)[N], int idx) {
  float r = v[0];
#pragma unroll
  for (int i = 1; i < N; ++i) r = (idx == i) ? v[i] : r;
  return r;
}

// DF inducing record, output columns b in [part*DQ, part*DQ + DQ) of NP column parts (DQ = ceil(D / NP)): the arithmetic of
// df_ind_half_bwd with the column picked at run time (`part` is wave-uniform; register arrays are read through select
// chains).  NP = 4 at D = 16 keeps the uniform-parameter partials at 2 x 16 x 4 registers instead of 2 x 16 x 8.
template <int D, int NP, bool WITH_G>
__device__ __forceinline__ void df_ind_part_bwd(const float4 (&r)[DfLayout<D>::RQ2], const float (&x)[D],
                                                const float* __restrict__ uni, int part, const float (&a)[D],
                                                float (&gx)[D], float (&g)[4 * DfLayout<D>::RQ2],
                                                float (&gwab)[D][(D + NP - 1) / NP], float (&gil2)[D][(D + NP - 1) / NP],
                                                float (&gvar)[(D + NP - 1) / NP]) {
  constexpr int DQ = (D + NP - 1) / NP;
  float f[4 * DfLayout<D>::RQ2];
  unpack(r, f);
  const float* wab = uni;
  const float* il2 = uni + D * D;
  const float* var = uni + 2 * D * D;
  float dl[D];
  float r2 = 0.f;
#pragma unroll
  for (int q = 0; q < D; ++q) { dl[q] = x[q] - f[q]; r2 = fmaf(dl[q], dl[q], r2); }
  float gd[D];
  float gr2 = 0.f;
#pragma unroll
  for (int q = 0; q < D; ++q) gd[q] = 0.f;
#pragma unroll
  for (int bb = 0; bb < DQ; ++bb) {
    const int b = part * DQ + bb;                    // wave-uniform
    const bool live = b < D;
    const int bidx = live ? b : D - 1;
    const float dlb = live ? pick<D>(dl, bidx) : 0.f;
    const float ab = live ? pick<D>(a, bidx) : 0.f;
    const float vb = var[bidx];
    float gdb = 0.f, gv = 0.f;
#pragma unroll
    for (int aa = 0; aa < D; ++aa) {
      const float il = il2[aa * D + bidx], wv = wab[aa * D + bidx];
      const float E = exp2_fast(r2 * wv);
      const bool diag = (aa == bidx);
      const float term = dl[aa] * dlb * il + (diag ? ((float)(D - 1) - r2 * il) : 0.f);
      const float G = ab * f[D + aa];
      const float GE = G * vb * E * il;
      gr2 = fmaf(GE, term * (GP_LN2 * wv) - (diag ? il : 0.f), gr2);
      gd[aa] = fmaf(GE * il, dlb, gd[aa]);
      gdb = fmaf(GE * il, dl[aa], gdb);
      if (WITH_G) {
        g[D + aa] = fmaf(ab, vb * E * il * term, g[D + aa]);
        gv = fmaf(G, E * il * term, gv);
        gwab[aa][bb] = fmaf(G * vb * il * term, E * GP_LN2 * r2, gwab[aa][bb]);
        gil2[aa][bb] = fmaf(G * vb * E, term + il * (dl[aa] * dlb - (diag ? r2 : 0.f)), gil2[aa][bb]);
      }
    }
    if (WITH_G) gvar[bb] += gv;
#pragma unroll
    for (int q = 0; q < D; ++q) gd[q] += (live && q == bidx) ? gdb : 0.f;
  }
#pragma unroll
  for (int q = 0; q < D; ++q) {
    const float v = fmaf(2.f * gr2, dl[q], gd[q]);
    gx[q] += v;
    if (WITH_G) g[q] -= v;
  }
}

template <int DI, int DO, int TS = TEAM> struct RbfStreamTeam {
  using L = RbfLayout<DI, DO>;
  static constexpr int kTeam = TS;                   // wavefronts per trajectory; 8 measured slower than 4 at D = 16 (256-VGPR cap: rollout 5.2 -> 8.1 ms)
  const float4* p4;
  const float4* i4;
  const float* wl;
  int nrec, nunit, wave, lane;
  TeamCombineT<TS> comb;
  __device__ __forceinline__ void init(const float* pack, int M, int S, float* lds, int w, int l) {
    p4 = reinterpret_cast<const float4*>(pack);
    i4 = p4 + L::rff_f4(S);
    wl = pack + 4 * (L::rff_f4(S) + L::ind_f4(M));
    nrec = cdiv(S, 64) * DO;
    nunit = 2 * cdiv(M, 64);
    wave = w; lane = l;
    comb.init(lds, w, l);
  }
  template <int MODE> __device__ __forceinline__ void eval(const float (&x)[DI], float (&f)[DO]) {
    float acc[DO];
#pragma unroll
    for (int d = 0; d < DO; ++d) acc[d] = 0.f;
    if (MODE != 2) {
      for (int rec = wave; rec < nrec; rec += TS) {
        float4 r[L::RQ];
        load_record<L::RQ>(p4, rec, lane, r);
        float v = 0.f;
        rbf_rff_record<DI, DO>(r, x, v);
        const int d = rec % DO;
#pragma unroll
        for (int i = 0; i < DO; ++i) acc[i] += (d == i) ? v : 0.f;
      }
    }
    if (MODE != 1) {
      for (int u = wave; u < nunit; u += TS) {
        asm volatile("" ::: "memory");
        float4 r[L::RQ2];
        load_record<L::RQ2>(i4, u >> 1, lane, r);
        rbf_ind_record_half<DI, DO>(r, x, wl, wave & 1, acc);
      }
    }
    float part[DO];
    wave_sum_all<DO>(acc, part);
    comb.template run<DO>(part, f);
  }
  __device__ __forceinline__ void vjp(const float (&x)[DI], const float (&a)[DO], float (&gx)[DI], bool prior_only = false) {
    float acc[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) acc[i] = 0.f;
    float g0[4 * L::RQ], g1[4 * L::RQ2], g2[(DO + 1) / 2][DI];
    for (int rec = wave; rec < nrec; rec += TS) {
      float4 r[L::RQ];
      load_record<L::RQ>(p4, rec, lane, r);
      rbf_rff_bwd<DI, DO, false>(r, x, pick<DO>(a, rec % DO), acc, g0);
    }
    if (!prior_only) {
      for (int u = wave; u < nunit; u += TS) {
        asm volatile("" ::: "memory");   // as below: uniform-table loads stay in the loop
        float4 r[L::RQ2];
        load_record<L::RQ2>(i4, u >> 1, lane, r);
        rbf_ind_half_bwd<DI, DO, false>(r, x, wl, wave & 1, a, acc, g1, g2);
      }
    }
    float part[DI];
    wave_sum_all<DI>(acc, part);
    comb.template run<DI>(part, gx);
  }
};

template <int D, int TS = TEAM> struct DfStreamTeam {
  using L = DfLayout<D>;
  static constexpr int kTeam = TS;
  static constexpr int NPB = D > 8 ? 4 : 2;          // column parts per inducing record in the backward (df_ind_part_bwd)
  static_assert(TS % NPB == 0 && TS % 2 == 0, "a wavefront keeps one column part for the whole launch");
  const float4* p4;
  const float4* i4;
  const float* uni;
  int nrec, nunit, wave, lane;
  TeamCombineT<TS> comb;
  __device__ __forceinline__ void init(const float* pack, int M, int S, float* lds, int w, int l) {
    p4 = reinterpret_cast<const float4*>(pack);
    i4 = p4 + L::rff_f4(S);
    uni = pack + 4 * (L::rff_f4(S) + L::ind_f4(M));
    nrec = cdiv(S, 64) * D;
    nunit = 2 * cdiv(M, 64);
    wave = w; lane = l;
    comb.init(lds, w, l);
  }
  template <int MODE> __device__ __forceinline__ void eval(const float (&x)[D], float (&f)[D]) {
    float acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) acc[d] = 0.f;
    if (MODE != 2) {
      for (int rec = wave; rec < nrec; rec += TS) {
        float4 r[L::RQ];
        load_record<L::RQ>(p4, rec, lane, r);
        df_rff_record<D>(r, x, acc);
      }
    }
    if (MODE != 1) {
      for (int u = wave; u < nunit; u += TS) {
        asm volatile("" ::: "memory");
        float4 r[L::RQ2];
        load_record<L::RQ2>(i4, u >> 1, lane, r);
        df_ind_record_half<D>(r, x, uni, wave & 1, acc);
      }
    }
    float part[D];
    wave_sum_all<D>(acc, part);
    comb.template run<D>(part, f);
  }
  __device__ __forceinline__ void vjp(const float (&x)[D], const float (&a)[D], float (&gx)[D], bool prior_only = false) {
    float acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = 0.f;
    float g0[4 * L::RQ], g1[4 * L::RQ2], g2[D][(D + NPB - 1) / NPB], g3[D][(D + NPB - 1) / NPB], g4[(D + NPB - 1) / NPB];
    for (int rec = wave; rec < nrec; rec += TS) {
      float4 r[L::RQ];
      load_record<L::RQ>(p4, rec, lane, r);
      df_rff_bwd<D, false>(r, x, a, acc, g0);
    }
    if (!prior_only) {
      for (int u = wave; u < NPB * (nunit / 2); u += TS) {   // unit u = NPB * record + column part
        asm volatile("" ::: "memory");   // keep the 2 D^2 + D uniform-table loads inside the loop (hoisted they spill)
        float4 r[L::RQ2];
        load_record<L::RQ2>(i4, u / NPB, lane, r);
        df_ind_part_bwd<D, NPB, false>(r, x, uni, wave % NPB, a, acc, g1, g2, g3, g4);
      }
    }
    float part[D];
    wave_sum_all<D>(acc, part);
    comb.template run<D>(part, gx);
  }
};


}  // namespace gp

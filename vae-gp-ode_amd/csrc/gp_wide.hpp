// gp_wide.hpp -- WIDE team evaluators: TS = 12 wavefronts per trajectory for the BASELINE latent widths (D_out in {6, 3}).
//
// Why.  A latent trajectory is a chain of 4 (T - 1) dependent right-hand-side evaluations.  With the 4-wavefront team of
// gp_team.hpp, batch 256 puts ONE wavefront on every SIMD: each wavefront then issues its ~260 VALU + ~30 transcendental
// instructions per evaluation back to back with nothing to hide their latencies behind (a lone wavefront sustains one vector
// instruction per 5-10 cycles in dependent code), and the evaluation ends in a 6-value wave reduction and a 24-read LDS combine.
// profiles/r01n: 3700 cycles per evaluation, 0.10 of the fp32 peak.  Here the same work is cut 12 ways, three wavefronts per
// SIMD, so that one wavefront's dependent chains run under the others' instructions:
//
//   work units of one evaluation (S <= 256, M <= 128):  4 D_out rff records (lane group j, dim), 2 D_out inducing units
//   (lane group j, output dim / column).  Wavefront w owns records w, w + 12, (...) and unit w -- its slice of the pack is
//   8-12 floats per lane and stays in registers for the whole launch.
//   RBF: 12 is a multiple of D_out, so everything a wavefront owns feeds ONE output dimension d = w mod D_out: its partial is a
//   single number (one-value wave reduction), and f_d is the sum of 12 / D_out slots.
//   DF: an rff record feeds every output (through B(omega)), so partials are D-vectors; the inducing unit is one output column.
//
// The combine is lane-parallel: wavefront w drops its partial vector into slot w (8 floats), ONE s_barrier, then every wavefront
// reads the 12 x 8 slot array with two ds_read_b32 per lane and adds it up with one DPP rotate and one permlane swap in a fixed
// order (bit-reproducible), six v_readlane deliver f.  Slots are double-buffered by evaluation parity as in TeamCombine.
#pragma once
#include "gp_team.hpp"

namespace gp {

template <int TS> struct WideCombine {
  static constexpr int DP = 8;                       // floats per wavefront slot
  static_assert(TS * DP <= 128 && TS % 4 == 0, "slot array read with two loads per lane");
  float* slots;                                      // [2][TS][DP] in LDS
  int wave, lane, parity;
  __device__ __forceinline__ void init(float* s, int w, int l) { slots = s; wave = w; lane = l; parity = 0; }
  // part: this wavefront's (wave-uniform) partial; f: the sum over the TS wavefronts, identical in every wavefront
  template <int NV> __device__ __forceinline__ void run(const float (&part)[NV], float (&f)[NV]) {
    static_assert(NV <= DP, "slot too small");
    float* base = slots + parity * TS * DP;
    if (lane < DP) {
      float v = 0.f;
#pragma unroll
      for (int d = 0; d < NV; ++d) v = (lane == d) ? part[d] : v;
      base[wave * DP + lane] = v;                    // the whole slot, zeros beyond NV
    }
    __syncthreads();
    // lane l holds element (w = l / 8, d = l % 8); elements 64 .. fold onto lanes 0 .. (same d)
    float v = base[lane];
    if (TS * DP > 64) v += (lane + 64 < TS * DP) ? base[(lane + 64 < TS * DP) ? lane + 64 : 0] : 0.f;
    v += dpp_mov<0x128>(v);                          // row_ror:8 -- + the other wavefront of the 16-lane row
    const float w2 = fold32(v, v);                   // lanes 0..31: v[l] + v[l + 32]
#pragma unroll
    for (int d = 0; d < NV; ++d) f[d] = GP_LANE(w2, d) + GP_LANE(w2, 16 + d);
    parity ^= 1;
  }
};

// one value summed over the 64 lanes of the wavefront (wave-uniform result)
__device__ __forceinline__ float wave_sum1(float x) {
  const float in1[1] = {x};
  float out1[1];
  wave_sum_multi<1>(in1, out1);
  return out1[0];
}

template <int DI, int DO, int TS> struct RbfWideTeam {
  using L = RbfLayout<DI, DO>;
  static constexpr int kTeam = TS;
  static_assert(TS % DO == 0, "a wavefront feeds one output dimension");
  static constexpr int NRW = (4 * DO + TS - 1) / TS;         // rff records per wavefront (S <= 256)
  static_assert((2 * DO + TS - 1) / TS == 1, "one inducing unit per wavefront (M <= 128)");
  float4 rff[NRW][L::RQ];
  float zz[DI], wld[DI], cc;
  float hot[DO];                                     // one-hot of this wavefront's output dimension (selects a runtime index
                                                     // without a register-indexed array: those go through scratch)
  int d;
  WideCombine<TS> comb;
  static __host__ bool fits(int M, int S) { return cdiv(S, 64) * DO <= NRW * TS && cdiv(M, 64) * DO <= TS; }
  __device__ __forceinline__ void init(const float* pack, int M, int S, float* lds, int wave, int lane) {
    const float4* p4 = reinterpret_cast<const float4*>(pack);
    const int SJ = cdiv(S, 64), MJ = cdiv(M, 64);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    wave = __builtin_amdgcn_readfirstlane(wave);     // wave-uniform by construction; tell the compiler (scalar selects below)
    d = wave % DO;
#pragma unroll
    for (int k = 0; k < NRW; ++k) {
      const int rec = wave + TS * k;                 // rec % DO == d: record (j, d), j = rec / DO
#pragma unroll
      for (int q = 0; q < L::RQ; ++q) rff[k][q] = rec < SJ * DO ? p4[((size_t)rec * L::RQ + q) * 64 + lane] : z;
    }
    const float4* i4 = p4 + L::rff_f4(S);
    const int j = wave / DO;                         // inducing unit (j, d)
    float4 ir[L::RQ2];
#pragma unroll
    for (int q = 0; q < L::RQ2; ++q) ir[q] = j < MJ ? i4[((size_t)j * L::RQ2 + q) * 64 + lane] : z;
    float f[4 * L::RQ2];
    unpack(ir, f);
#pragma unroll
    for (int i = 0; i < DI; ++i) zz[i] = f[i];
    cc = 0.f;
#pragma unroll
    for (int dd = 0; dd < DO; ++dd) cc = (dd == d) ? f[DI + dd] : cc;
    const float* wl = pack + 4 * (L::rff_f4(S) + L::ind_f4(M));
#pragma unroll
    for (int i = 0; i < DI; ++i) wld[i] = wl[d * DI + i];
#pragma unroll
    for (int dd = 0; dd < DO; ++dd) hot[dd] = (dd == d) ? 1.f : 0.f;
    comb.init(lds, wave, lane);
  }
  template <int MODE> __device__ __forceinline__ void eval(const float (&x)[DI], float (&f)[DO]) {
    float acc = 0.f;
    if (MODE != 2) {
#pragma unroll
      for (int k = 0; k < NRW; ++k) rbf_rff_record<DI, DO>(rff[k], x, acc);
    }
    if (MODE != 1) {
      float e = 0.f;
#pragma unroll
      for (int i = 0; i < DI; ++i) { const float dl = x[i] - zz[i]; e = fmaf(wld[i], dl * dl, e); }
      acc = fmaf(cc, exp2_fast(e), acc);
    }
    const float s = wave_sum1(acc);
    float part[DO];
#pragma unroll
    for (int dd = 0; dd < DO; ++dd) part[dd] = hot[dd] * s;
    comb.template run<DO>(part, f);
  }
  __device__ __forceinline__ void vjp(const float (&x)[DI], const float (&a)[DO], float (&gx)[DI], bool prior_only = false) {
    float acc[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) acc[i] = 0.f;
    float ad = 0.f;
#pragma unroll
    for (int dd = 0; dd < DO; ++dd) ad = fmaf(hot[dd], a[dd], ad);
    float g0[4 * L::RQ];
#pragma unroll
    for (int k = 0; k < NRW; ++k) rbf_rff_bwd<DI, DO, false>(rff[k], x, ad, acc, g0);
    if (!prior_only) {
      float dl[DI], e = 0.f;
#pragma unroll
      for (int i = 0; i < DI; ++i) { dl[i] = x[i] - zz[i]; e = fmaf(wld[i], dl[i] * dl[i], e); }
      const float w = ad * exp2_fast(e) * cc * GP_LN2;       // d L / d e
#pragma unroll
      for (int i = 0; i < DI; ++i) acc[i] = fmaf(w * wld[i], 2.f * dl[i], acc[i]);
    }
    float part[DI];
    wave_sum_all<DI>(acc, part);
    comb.template run<DI>(part, gx);
  }
};

// DF inducing record, ONE output column b (wave-uniform): the arithmetic of df_ind_record for that column, with the column's
// uniform parameters (il2[a][b], wab[a][b], var[b]) already in registers
template <int D>
__device__ __forceinline__ float df_ind_record_col(const float (&zz)[D], const float (&nn)[D], const float (&x)[D], const float (&ilb)[D],
                                                   const float (&wab)[D], float varb, const float (&hot)[D]) {
  float dl[D];
  float r2 = 0.f, dlb = 0.f;
#pragma unroll
  for (int a = 0; a < D; ++a) { dl[a] = x[a] - zz[a]; r2 = fmaf(dl[a], dl[a], r2); dlb = fmaf(hot[a], dl[a], dlb); }
  float sb = 0.f;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    const float il = ilb[a];
    const float E = exp2_fast(r2 * wab[a]);
    float term = dl[a] * dlb * il;
    term = fmaf(hot[a], (float)(D - 1) - r2 * il, term);   // the diagonal term of column b, on row a == b only
    sb = fmaf(nn[a] * (E * il), term, sb);
  }
  return varb * sb;
}

// d/dx of the same column (the arithmetic of df_ind_half_bwd / df_ind_part_bwd restricted to one column, no parameter gradients)
template <int D>
__device__ __forceinline__ void df_ind_col_vjp(const float (&zz)[D], const float (&nn)[D], const float (&x)[D], const float (&ilb)[D],
                                               const float (&wab)[D], float varb, const float (&hot)[D], const float (&a)[D],
                                               float (&gx)[D]) {
  float dl[D];
  float r2 = 0.f, dlb = 0.f, ab = 0.f;
#pragma unroll
  for (int q = 0; q < D; ++q) { dl[q] = x[q] - zz[q]; r2 = fmaf(dl[q], dl[q], r2); dlb = fmaf(hot[q], dl[q], dlb); ab = fmaf(hot[q], a[q], ab); }
  float gd[D];
  float gr2 = 0.f, gdb = 0.f;
#pragma unroll
  for (int aa = 0; aa < D; ++aa) {
    const float il = ilb[aa], wv = wab[aa];
    const float E = exp2_fast(r2 * wv);
    const float term = fmaf(hot[aa], (float)(D - 1) - r2 * il, dl[aa] * dlb * il);
    const float GE = ab * nn[aa] * varb * E * il;
    gr2 = fmaf(GE, term * (GP_LN2 * wv) - hot[aa] * il, gr2);
    gd[aa] = GE * il * dlb;
    gdb = fmaf(GE * il, dl[aa], gdb);
  }
#pragma unroll
  for (int q = 0; q < D; ++q) gx[q] += fmaf(2.f * gr2, dl[q], fmaf(hot[q], gdb, gd[q]));
}

template <int D, int TS> struct DfWideTeam {
  using L = DfLayout<D>;
  static constexpr int kTeam = TS;
  static constexpr int NRW = (4 * D + TS - 1) / TS;          // rff records per wavefront (S <= 256)
  static_assert((2 * D + TS - 1) / TS == 1, "one inducing unit (lane group, column) per wavefront (M <= 128)");
  float4 rff[NRW][L::RQ];
  float4 ind[L::RQ2];
  float zz[D], nn[D], ilb[D], wabb[D], varb, hot[D];
  const float* uni;
  int b, unit_ok;
  WideCombine<TS> comb;
  static __host__ bool fits(int M, int S) { return cdiv(S, 64) * D <= NRW * TS && cdiv(M, 64) * D <= TS; }
  __device__ __forceinline__ void init(const float* pack, int M, int S, float* lds, int wave, int lane) {
    const float4* p4 = reinterpret_cast<const float4*>(pack);
    const int SJ = cdiv(S, 64), MJ = cdiv(M, 64);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < NRW; ++k) {
      const int rec = wave + TS * k;
#pragma unroll
      for (int q = 0; q < L::RQ; ++q) rff[k][q] = rec < SJ * D ? p4[((size_t)rec * L::RQ + q) * 64 + lane] : z;
    }
    wave = __builtin_amdgcn_readfirstlane(wave);     // wave-uniform by construction; tell the compiler (scalar selects below)
    const float4* i4 = p4 + L::rff_f4(S);
    const int j = wave / D;                          // unit (j, b): lane group j, output column b
    b = wave % D;
    unit_ok = j < MJ;
#pragma unroll
    for (int q = 0; q < L::RQ2; ++q) ind[q] = unit_ok ? i4[((size_t)j * L::RQ2 + q) * 64 + lane] : z;
    float f[4 * L::RQ2];
    unpack(ind, f);
#pragma unroll
    for (int a = 0; a < D; ++a) { zz[a] = f[a]; nn[a] = f[D + a]; }
    uni = pack + 4 * (L::rff_f4(S) + L::ind_f4(M));
#pragma unroll
    for (int a = 0; a < D; ++a) { wabb[a] = uni[a * D + b]; ilb[a] = uni[D * D + a * D + b]; }
    varb = uni[2 * D * D + b];
#pragma unroll
    for (int a = 0; a < D; ++a) hot[a] = (a == b) ? 1.f : 0.f;
    comb.init(lds, wave, lane);
  }
  template <int MODE> __device__ __forceinline__ void eval(const float (&x)[D], float (&f)[D]) {
    float acc[D];
#pragma unroll
    for (int q = 0; q < D; ++q) acc[q] = 0.f;
    if (MODE != 2) {
#pragma unroll
      for (int k = 0; k < NRW; ++k) df_rff_record<D>(rff[k], x, acc);
    }
    if (MODE != 1) {
      const float v = df_ind_record_col<D>(zz, nn, x, ilb, wabb, varb, hot);   // zero coefficients in padding lanes / missing units
#pragma unroll
      for (int q = 0; q < D; ++q) acc[q] = fmaf(hot[q], v, acc[q]);
    }
    float part[D];
    wave_sum_all<D>(acc, part);
    comb.template run<D>(part, f);
  }
  __device__ __forceinline__ void vjp(const float (&x)[D], const float (&a)[D], float (&gx)[D], bool prior_only = false) {
    float acc[D];
#pragma unroll
    for (int i = 0; i < D; ++i) acc[i] = 0.f;
    float g0[4 * L::RQ];
#pragma unroll
    for (int k = 0; k < NRW; ++k) df_rff_bwd<D, false>(rff[k], x, a, acc, g0);
    if (!prior_only) df_ind_col_vjp<D>(zz, nn, x, ilb, wabb, varb, hot, a, acc);
    float part[D];
    wave_sum_all<D>(acc, part);
    comb.template run<D>(part, gx);
  }
};

// Measured (MI355X, rk4, T = 16; rollout launch, us): the wide team LOSES to the 4-wavefront team at every BASELINE shape --
//   configs[0] RBF batch 32: 53 vs 48   configs[3] RBF batch 256: 74 vs 67   configs[1] DF batch 256: 105 vs 94   configs[2]: 72 vs 58
// An evaluation's arithmetic does shrink (~850 -> ~280 issue cycles per wavefront), but it was never the long pole: the chain
// reduce -> LDS slot -> s_barrier -> LDS read -> cross-lane sum -> v_readlane is ~1500 cycles of pure latency per evaluation either
// way, and twelve wavefronts reach the barrier with more skew than four.  So it stays an A/B switch (GPODE_TEAM_WIDE=1) and the
// parity tests cover it (tests/test_gpu_forward.py::test_wide_team_matches); the default is the 4-wavefront team.
static constexpr int kWideMaxRows = 512;
static inline bool wide_team_enabled() {
  static const bool on = [] { const char* e = getenv("GPODE_TEAM_WIDE"); return e && e[0] == '1'; }();
  return on;
}

}  // namespace gp

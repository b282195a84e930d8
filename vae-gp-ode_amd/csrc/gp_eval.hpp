// gp_eval.hpp -- device evaluators of the sparse-GP vector field f(x) for CDNA4 (gfx950).
//
// Mapping ("wave per trajectory"): one 64-lane wavefront owns one latent state x (replicated in
// every lane).  The RFF feature index s and the inducing index m are spread over the 64 lanes;
// each lane accumulates its partial f[0..Do) in registers and one butterfly all-reduce per
// evaluation leaves f in every lane, so the Runge-Kutta stage algebra needs no LDS and no barrier.
//
// Packed per-draw cache ("pack", written by gp_cache.hip): lane-major *records* of float4,
//     float4 index = (rec * RQ + q) * 64 + lane
// so a wave reads 1 KiB per instruction (global_load_dwordx4 / ds_read_b128, conflict-free).
//
//  RBF (kernels.py:140-153,174-181), SJ = ceil(S/64), MJ = ceil(M/64):
//    rff record (j,d), s = 64 j + lane : [ om[0..Di) = eps[i,s,d]/(ell[d,i] 2 pi), ph = u[s,d],
//                                          aw = sqrt(var_d/S) w[s,d] ]           RQ  = ceil((Di+2)/4)
//    ind record j,     m = 64 j + lane : [ zz[0..Di) = Z[m,i], cc[0..Do) = var_d nu[d,m] ]
//                                                                                RQ2 = ceil((Di+Do)/4)
//    uniform tail: wl[d][i] = -0.5 log2(e) / ell[d,i]^2
//  DF (kernels.py:289-303,319-351,390-393), D = Di = Do:
//    rff record (j,i), s = 64 j + lane : [ om[0..D) = omega[k,s,i]/(2 pi), ph = u[s,i], wc = w[s,i],
//                                          ws = w[S+s,i], bs[0..D) = B[s,i,jj] sqrt(var_jj/S) ]
//                                                                                RQ  = ceil((2D+3)/4)
//    ind record j,     m = 64 j + lane : [ zz[0..D) = Z[m,a], nn[0..D) = nu[(m,a)] ]   RQ2 = ceil(2D/4)
//    uniform tail: wab[a][b] = -log2(e)/(2 ell[a,b]^2), il2[a][b] = 1/ell[a,b]^2, var[b]
//  Lanes with s >= S or m >= M hold zeros in aw / wc,ws / cc / nn, which masks them for free.
//
// Numerics: cos/sin are evaluated in revolutions (v_fract + v_cos/v_sin: the phase 2*pi*u and the
// 1/(2 pi) of omega are folded into the pack), exp as v_exp_f32 (2^x) with log2(e) folded into the
// lengthscale weights, and the squared distance in difference form (the reference uses the
// expanded form, kernels.py:64-79, which loses ~1e-6 to cancellation).  fp32 throughout.
#pragma once
#include <hip/hip_runtime.h>
#include "wave_reduce.hpp"

#define GP_WAVE 64
#define GP_LOG2E 1.4426950408889634f
#define GP_INV2PI 0.15915494309189535f

namespace gp {

__host__ __device__ constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

template <int DI, int DO> struct RbfLayout {
  static constexpr int RQ = cdiv(DI + 2, 4);
  static constexpr int RQ2 = cdiv(DI + DO, 4);
  __host__ __device__ static size_t rff_f4(int S) { return (size_t)cdiv(S, 64) * DO * RQ * 64; }
  __host__ __device__ static size_t ind_f4(int M) { return (size_t)cdiv(M, 64) * RQ2 * 64; }
  __host__ __device__ static size_t uni_floats() { return (size_t)cdiv(DO * DI, 4) * 4; }
  __host__ __device__ static size_t total_floats(int M, int S) { return 4 * (rff_f4(S) + ind_f4(M)) + uni_floats(); }
};

template <int D> struct DfLayout {
  static constexpr int RQ = cdiv(2 * D + 3, 4);
  static constexpr int RQ2 = cdiv(2 * D, 4);
  __host__ __device__ static size_t rff_f4(int S) { return (size_t)cdiv(S, 64) * D * RQ * 64; }
  __host__ __device__ static size_t ind_f4(int M) { return (size_t)cdiv(M, 64) * RQ2 * 64; }
  __host__ __device__ static size_t uni_floats() { return (size_t)cdiv(2 * D * D + D, 4) * 4; }
  __host__ __device__ static size_t total_floats(int M, int S) { return 4 * (rff_f4(S) + ind_f4(M)) + uni_floats(); }
};

__device__ __forceinline__ float wave_allreduce_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// cos / sin of 2*pi*t, t in revolutions (any magnitude the fp32 fract can resolve)
__device__ __forceinline__ float cos_rev(float t) { return __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(t)); }
__device__ __forceinline__ float sin_rev(float t) { return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(t)); }
__device__ __forceinline__ float exp2_fast(float t) { return __builtin_amdgcn_exp2f(t); }

template <int NQ> __device__ __forceinline__ void unpack(const float4 (&r)[NQ], float (&f)[4 * NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    f[4 * q + 0] = r[q].x; f[4 * q + 1] = r[q].y; f[4 * q + 2] = r[q].z; f[4 * q + 3] = r[q].w;
  }
}

// ----------------------------------------------------------------------------------------------
// RBF: one rff record contributes to output d; one ind record contributes to every d.
// ----------------------------------------------------------------------------------------------
template <int DI, int DO>
__device__ __forceinline__ void rbf_rff_record(const float4 (&r)[RbfLayout<DI, DO>::RQ], const float (&x)[DI], float& acc) {
  float f[4 * RbfLayout<DI, DO>::RQ];
  unpack(r, f);
  float t = f[DI];
#pragma unroll
  for (int i = 0; i < DI; ++i) t = fmaf(x[i], f[i], t);
  acc = fmaf(f[DI + 1], cos_rev(t), acc);
}

template <int DI, int DO>
__device__ __forceinline__ void rbf_ind_record(const float4 (&r)[RbfLayout<DI, DO>::RQ2], const float (&x)[DI],
                                               const float* __restrict__ wl, float (&acc)[DO]) {
  float f[4 * RbfLayout<DI, DO>::RQ2];
  unpack(r, f);
  float t2[DI];
#pragma unroll
  for (int i = 0; i < DI; ++i) { float d = x[i] - f[i]; t2[i] = d * d; }
#pragma unroll
  for (int d = 0; d < DO; ++d) {
    float e = 0.f;
#pragma unroll
    for (int i = 0; i < DI; ++i) e = fmaf(wl[d * DI + i], t2[i], e);
    acc[d] = fmaf(f[DI + d], exp2_fast(e), acc[d]);
  }
}

// ----------------------------------------------------------------------------------------------
// DF
// ----------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void df_rff_record(const float4 (&r)[DfLayout<D>::RQ], const float (&x)[D], float (&acc)[D]) {
  float f[4 * DfLayout<D>::RQ];
  unpack(r, f);
  float t = f[D];
#pragma unroll
  for (int k = 0; k < D; ++k) t = fmaf(x[k], f[k], t);
  t = __builtin_amdgcn_fractf(t);
  float c = __builtin_amdgcn_cosf(t), s = __builtin_amdgcn_sinf(t);
  float rr = fmaf(f[D + 2], s, f[D + 1] * c);
#pragma unroll
  for (int j = 0; j < D; ++j) acc[j] = fmaf(rr, f[D + 3 + j], acc[j]);
}

template <int D>
__device__ __forceinline__ void df_ind_record(const float4 (&r)[DfLayout<D>::RQ2], const float (&x)[D],
                                              const float* __restrict__ uni, float (&acc)[D]) {
  float f[4 * DfLayout<D>::RQ2];
  unpack(r, f);
  const float* wab = uni;
  const float* il2 = uni + D * D;
  const float* var = uni + 2 * D * D;
  float dl[D];
  float r2 = 0.f;
#pragma unroll
  for (int a = 0; a < D; ++a) { dl[a] = x[a] - f[a]; r2 = fmaf(dl[a], dl[a], r2); }
#pragma unroll
  for (int b = 0; b < D; ++b) {
    float sb = 0.f;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      float il = il2[a * D + b];
      float E = exp2_fast(r2 * wab[a * D + b]);
      float term = dl[a] * dl[b] * il;
      if (a == b) term += (float)(D - 1) - r2 * il;
      sb = fmaf(f[D + a] * (E * il), term, sb);
    }
    acc[b] = fmaf(var[b], sb, acc[b]);
  }
}

// ----------------------------------------------------------------------------------------------
// Half-range variants for the 4-wave team mapping: one wave handles the output dims
// [half*DH, min(DO,(half+1)*DH)) of one inducing record, DH = ceil(DO/2).  `half` is wave-uniform;
// every register index below is a compile-time constant (runtime choices are selects), so nothing
// is demoted to scratch.  Results are merged into acc[] at their global positions.
// ----------------------------------------------------------------------------------------------
template <int DI, int DO>
__device__ __forceinline__ void rbf_ind_record_half(const float4 (&r)[RbfLayout<DI, DO>::RQ2], const float (&x)[DI],
                                                    const float* __restrict__ wl, int half, float (&acc)[DO]) {
  constexpr int DH = (DO + 1) / 2;
  float f[4 * RbfLayout<DI, DO>::RQ2];
  unpack(r, f);
  float t2[DI];
#pragma unroll
  for (int i = 0; i < DI; ++i) { float d = x[i] - f[i]; t2[i] = d * d; }
#pragma unroll
  for (int dd = 0; dd < DH; ++dd) {
    // wave-uniform row of wl (scalar loads); the upper half of an odd DO has one row less: clamp, its
    // coefficient is zero anyway but the exponent must stay finite (0 * inf = NaN)
    const float* wlh = wl + ((half * DH + dd < DO) ? half * DH + dd : 0) * DI;
    float e = 0.f;
#pragma unroll
    for (int i = 0; i < DI; ++i) e = fmaf(wlh[i], t2[i], e);
    const float c_lo = f[DI + dd];
    const float c_hi = (DH + dd < DO) ? f[DI + (DH + dd < DO ? DH + dd : 0)] : 0.f;
    const float v = (half ? c_hi : c_lo) * exp2_fast(e);
    acc[dd] += half ? 0.f : v;
    if (DH + dd < DO) acc[DH + dd < DO ? DH + dd : 0] += half ? v : 0.f;
  }
}

template <int D>
__device__ __forceinline__ void df_ind_record_half(const float4 (&r)[DfLayout<D>::RQ2], const float (&x)[D],
                                                   const float* __restrict__ uni, int half, float (&acc)[D]) {
  constexpr int DH = (D + 1) / 2;
  float f[4 * DfLayout<D>::RQ2];
  unpack(r, f);
  const float* wab = uni;
  const float* il2 = uni + D * D;
  const float* var = uni + 2 * D * D;
  float dl[D];
  float r2 = 0.f;
#pragma unroll
  for (int a = 0; a < D; ++a) { dl[a] = x[a] - f[a]; r2 = fmaf(dl[a], dl[a], r2); }
  const int b0 = half * DH;  // wave-uniform
#pragma unroll
  for (int bb = 0; bb < DH; ++bb) {
    const bool valid = (DH + bb < D);                     // does the upper half have this column?
    const int bidx = b0 + bb < D ? b0 + bb : D - 1;       // uniform index for the scalar loads
    const float dlb = half ? (valid ? dl[DH + bb < D ? DH + bb : 0] : 0.f) : dl[bb];
    float sb = 0.f;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const float il = il2[a * D + bidx];
      const float E = exp2_fast(r2 * wab[a * D + bidx]);
      float term = dl[a] * dlb * il;
      term += (a == bidx) ? ((float)(D - 1) - r2 * il) : 0.f;
      sb = fmaf(f[D + a] * (E * il), term, sb);
    }
    const float v = var[bidx] * sb;
    acc[bb] += half ? 0.f : v;
    if (valid) acc[DH + bb < D ? DH + bb : 0] += half ? v : 0.f;
  }
}

// ==============================================================================================
// Backward building blocks.  `a` is the (wave-uniform) upstream adjoint of f; gx accumulates J^T a;
// g* accumulate the gradient w.r.t. the fields of the record, in the record's own float layout.
// d/dt cos(2 pi t) = -2 pi sin(2 pi t): the 2 pi of "revolutions" shows up in every chain rule below.
// ==============================================================================================
#define GP_2PI 6.283185307179586f
#define GP_LN2 0.6931471805599453f

// RBF rff record for output d.  WITH_G: also accumulate g[i] (om_i) and g[DI+1] (aw).
template <int DI, int DO, bool WITH_G>
__device__ __forceinline__ void rbf_rff_bwd(const float4 (&r)[RbfLayout<DI, DO>::RQ], const float (&x)[DI], float a_d,
                                            float (&gx)[DI], float (&g)[4 * RbfLayout<DI, DO>::RQ]) {
  float f[4 * RbfLayout<DI, DO>::RQ];
  unpack(r, f);
  float t = f[DI];
#pragma unroll
  for (int i = 0; i < DI; ++i) t = fmaf(x[i], f[i], t);
  t = __builtin_amdgcn_fractf(t);
  const float sn = __builtin_amdgcn_sinf(t);
  const float coef = a_d * f[DI + 1] * (-GP_2PI) * sn;  // a_d * d f_d / d t
#pragma unroll
  for (int i = 0; i < DI; ++i) gx[i] = fmaf(coef, f[i], gx[i]);
  if (WITH_G) {
    const float c = __builtin_amdgcn_cosf(t);
#pragma unroll
    for (int i = 0; i < DI; ++i) g[i] = fmaf(coef, x[i], g[i]);
    g[DI + 1] = fmaf(a_d, c, g[DI + 1]);
  }
}

// RBF inducing record, output dims of one half.  gwl[dd][i]: per-lane partial of d/d wl[half*DH+dd][i].
template <int DI, int DO, bool WITH_G>
__device__ __forceinline__ void rbf_ind_half_bwd(const float4 (&r)[RbfLayout<DI, DO>::RQ2], const float (&x)[DI],
                                                 const float* __restrict__ wl, int half, const float (&a)[DO],
                                                 float (&gx)[DI], float (&g)[4 * RbfLayout<DI, DO>::RQ2],
                                                 float (&gwl)[(DO + 1) / 2][DI]) {
  constexpr int DH = (DO + 1) / 2;
  float f[4 * RbfLayout<DI, DO>::RQ2];
  unpack(r, f);
  float dl[DI], t2[DI];
#pragma unroll
  for (int i = 0; i < DI; ++i) { dl[i] = x[i] - f[i]; t2[i] = dl[i] * dl[i]; }
  float gxi[DI];
#pragma unroll
  for (int i = 0; i < DI; ++i) gxi[i] = 0.f;
#pragma unroll
  for (int dd = 0; dd < DH; ++dd) {
    const bool hi_ok = DH + dd < DO;
    const float* wlh = wl + ((half * DH + dd < DO) ? half * DH + dd : 0) * DI;  // clamped, see rbf_ind_record_half
    float e = 0.f;
#pragma unroll
    for (int i = 0; i < DI; ++i) e = fmaf(wlh[i], t2[i], e);
    const float E = exp2_fast(e);
    const float c_lo = f[DI + dd], c_hi = hi_ok ? f[DI + (hi_ok ? DH + dd : 0)] : 0.f;
    const float a_lo = a[dd], a_hi = hi_ok ? a[hi_ok ? DH + dd : 0] : 0.f;
    const float cc = half ? c_hi : c_lo, ad = half ? a_hi : a_lo;
    const float aE = ad * E;           // d L / d cc
    const float w = aE * cc * GP_LN2;  // d L / d e
#pragma unroll
    for (int i = 0; i < DI; ++i) gxi[i] = fmaf(w * wlh[i], 2.f * dl[i], gxi[i]);
    if (WITH_G) {
      g[DI + dd] += half ? 0.f : aE;
      if (hi_ok) g[DI + (hi_ok ? DH + dd : 0)] += half ? aE : 0.f;
#pragma unroll
      for (int i = 0; i < DI; ++i) gwl[dd][i] = fmaf(w, t2[i], gwl[dd][i]);
    }
  }
#pragma unroll
  for (int i = 0; i < DI; ++i) {
    gx[i] += gxi[i];
    if (WITH_G) g[i] -= gxi[i];  // d/dz = -d/dx
  }
}

// DF rff record (s,i).  WITH_G: g[k] (om_k), g[D+3+j] (bs_j).
template <int D, bool WITH_G>
__device__ __forceinline__ void df_rff_bwd(const float4 (&r)[DfLayout<D>::RQ], const float (&x)[D], const float (&a)[D],
                                           float (&gx)[D], float (&g)[4 * DfLayout<D>::RQ]) {
  float f[4 * DfLayout<D>::RQ];
  unpack(r, f);
  float t = f[D];
#pragma unroll
  for (int k = 0; k < D; ++k) t = fmaf(x[k], f[k], t);
  t = __builtin_amdgcn_fractf(t);
  const float c = __builtin_amdgcn_cosf(t), sn = __builtin_amdgcn_sinf(t);
  float A = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) A = fmaf(a[j], f[D + 3 + j], A);
  const float rt = GP_2PI * fmaf(f[D + 2], c, -f[D + 1] * sn);  // d rr / d t
  const float coef = A * rt;
#pragma unroll
  for (int k = 0; k < D; ++k) gx[k] = fmaf(coef, f[k], gx[k]);
  if (WITH_G) {
    const float rr = fmaf(f[D + 2], sn, f[D + 1] * c);
#pragma unroll
    for (int k = 0; k < D; ++k) g[k] = fmaf(coef, x[k], g[k]);
#pragma unroll
    for (int j = 0; j < D; ++j) g[D + 3 + j] = fmaf(a[j], rr, g[D + 3 + j]);
  }
}

// DF inducing record, output columns b of one half.  Uniform-parameter partials (per lane):
//   gwab[a][bb], gil2[a][bb] for b = half*DH + bb, gvar[bb].
template <int D, bool WITH_G>
__device__ __forceinline__ void df_ind_half_bwd(const float4 (&r)[DfLayout<D>::RQ2], const float (&x)[D],
                                                const float* __restrict__ uni, int half, const float (&a)[D],
                                                float (&gx)[D], float (&g)[4 * DfLayout<D>::RQ2],
                                                float (&gwab)[D][(D + 1) / 2], float (&gil2)[D][(D + 1) / 2],
                                                float (&gvar)[(D + 1) / 2]) {
  constexpr int DH = (D + 1) / 2;
  float f[4 * DfLayout<D>::RQ2];
  unpack(r, f);
  const float* wab = uni;
  const float* il2 = uni + D * D;
  const float* var = uni + 2 * D * D;
  float dl[D];
  float r2 = 0.f;
#pragma unroll
  for (int q = 0; q < D; ++q) { dl[q] = x[q] - f[q]; r2 = fmaf(dl[q], dl[q], r2); }
  const int b0 = half * DH;
  float gd[D];      // d L / d delta_c
  float gr2 = 0.f;  // d L / d r2 (through E and the diagonal term), applied as 2 delta_c at the end
#pragma unroll
  for (int q = 0; q < D; ++q) gd[q] = 0.f;
#pragma unroll
  for (int bb = 0; bb < DH; ++bb) {
    const bool hi_ok = DH + bb < D;
    const int bidx = b0 + bb < D ? b0 + bb : D - 1;  // wave-uniform
    const bool live = half ? hi_ok : true;
    const float dlb = half ? (hi_ok ? dl[hi_ok ? DH + bb : 0] : 0.f) : dl[bb];
    const float ab = live ? (half ? (hi_ok ? a[hi_ok ? DH + bb : 0] : 0.f) : a[bb]) : 0.f;
    const float vb = var[bidx];
    float gdb = 0.f;  // d L / d delta_b from the cross term
    float gv = 0.f;
#pragma unroll
    for (int aa = 0; aa < D; ++aa) {
      const float il = il2[aa * D + bidx], wv = wab[aa * D + bidx];
      const float E = exp2_fast(r2 * wv);
      const bool diag = (aa == bidx);
      const float term = dl[aa] * dlb * il + (diag ? ((float)(D - 1) - r2 * il) : 0.f);
      const float G = ab * f[D + aa];               // upstream x coefficient nu[(m,aa)]
      const float GE = G * vb * E * il;             // G * dT/dterm
      // T = vb * E * il * term
      gr2 = fmaf(GE, term * (GP_LN2 * wv) - (diag ? il : 0.f), gr2);
      gd[aa] = fmaf(GE * il, dlb, gd[aa]);
      gdb = fmaf(GE * il, dl[aa], gdb);
      if (WITH_G) {
        g[D + aa] = fmaf(ab, vb * E * il * term, g[D + aa]);                       // d/d nu[(m,aa)]
        gv = fmaf(G, E * il * term, gv);                                           // d/d var_b
        gwab[aa][bb] = fmaf(G * vb * il * term, E * GP_LN2 * r2, gwab[aa][bb]);    // d/d wab[aa][b]
        gil2[aa][bb] = fmaf(G * vb * E, term + il * (dl[aa] * dlb - (diag ? r2 : 0.f)), gil2[aa][bb]);
      }
    }
    if (WITH_G) gvar[bb] += gv;
    // scatter d/d delta_b to its compile-time slot
    if (live) {
      if (half) { if (hi_ok) gd[hi_ok ? DH + bb : 0] += gdb; }
      else gd[bb] += gdb;
    }
  }
#pragma unroll
  for (int q = 0; q < D; ++q) {
    const float v = fmaf(2.f * gr2, dl[q], gd[q]);
    gx[q] += v;
    if (WITH_G) g[q] -= v;
  }
}

}  // namespace gp

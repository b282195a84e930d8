// gp_cache.hip -- per-draw cache build (forward) for the sparse-GP vector field.
//
// Replaces SVGP_Layer.build_cache (svpy.py:103-121):
//   kern.build_cache        kernels.py:126-137 / :305-316   omega = eps/ell, phase = 2 pi u
//   sample_inducing         svpy.py:88-101                  u = tril(Us) eps_u + Um
//   kern.K(Z)               kernels.py:98-110 / :289-303
//   kern.rff_forward(Z)     kernels.py:140-153 / :319-351   (via the rhs kernel, prior-only mode)
//   kern.compute_nu         kernels.py:155-172 / :376-387   nu = L^-T (u - L^-1 f_prior(Z))
// and writes the lane-major pack documented in gp_eval.hpp.
//
// Linear algebra: blocked left-looking Cholesky, NB = 32, ONE launch per block column, batched over
// gridDim.y (RBF: one M x M system per output dim; DF: a single (M D)^2 system).  Every workgroup of
// a launch owns one 32-row block of the panel; it forms its update with LDS-tiled FMAs, refactors the
// 32x32 diagonal block redundantly in the registers of one wavefront (v_readlane broadcasts, no
// barriers), inverts it, and applies the inverse to its block.  The right-hand side f_prior(Z) is
// appended as row n of the matrix, so the forward substitution L^-1 f_prior(Z) falls out of the
// panel solves for free (row n of the factor); only the transposed solve runs as its own kernel.
#include <type_traits>
#include "gp_eval.hpp"
#include "gp_launch.hpp"
#include "wave_reduce.hpp"
#include "gp_ws.hpp"

namespace gp {

static constexpr float kJitter = 1e-5f;  // kernels.py:11

__device__ __forceinline__ float softplus_lower(float x) {
  // F.softplus(x) + 1e-12  (constraint_utils.py:5-7; torch threshold 20)
  float sp = x > 20.f ? x : log1pf(expf(x));
  return sp + 1e-12f;
}

// ---------------------------------------------------------------------------------------------
// k_prep: everything that depends only on the raw parameters and the noise, in ONE launch.
// Block ranges (256 threads each):
//   [0, nb_rff)            rff records of the pack            (thread per (j, d|i, lane))
//   [.., + nb_u)           inducing sample u = tril(Us) eps_u + Um   (one wavefront per (n,d) row)
//   [.., + nb_ind)         inducing records: Z, zeroed coefficient fields
//   [.., + nb_hyp)         ell, var (softplus + 1e-12), uniform tail of the pack
//   [.., + nb_om)          API-visible omega (Di,S,Do) and phase (1,S,Do)
// ---------------------------------------------------------------------------------------------
struct PrepArgs {
  int kernel, Di, Do, M, S;
  const float *raw_ell, *raw_var, *Z, *Um, *Us, *eps_u, *rff_w, *rff_eps, *rff_u;
  float *pack, *pack_ind, *uni;
  float *ell_ws, *var_ws, *u_ws;
  float *ell_out, *var_out, *omega_out, *phase_out, *u_out;
  int* info;                                         // status word of the factorisation, cleared here (no memset node)
  int nb_rff, nb_u, nb_ind, nb_hyp, nb_om;
  // blockIdx.y = Monte-Carlo draw: per-draw strides (floats) of the noise, of the pack and of the per-draw outputs
  size_t s_eps_u, s_rff_w, s_rff_eps, s_rff_u, s_pack;
};

__device__ __forceinline__ void put_rec(float* __restrict__ pack, size_t rec_f4_base, int lane, int field, float v) {
  // float index of (record base in float4 units, quad q = field/4, lane, component field%4)
  pack[((rec_f4_base + (size_t)(field >> 2)) * 64 + lane) * 4 + (field & 3)] = v;
}

__global__ __launch_bounds__(256) void k_prep(PrepArgs a) {
  const int Di = a.Di, Do = a.Do, S = a.S, M = a.M;
  {
    const size_t dr = blockIdx.y;                    // this draw's noise, pack and outputs (ell / var / info: same values from every draw)
    a.eps_u += dr * a.s_eps_u; a.rff_w += dr * a.s_rff_w; a.rff_eps += dr * a.s_rff_eps; a.rff_u += dr * a.s_rff_u;
    a.pack += dr * a.s_pack; a.pack_ind += dr * a.s_pack; a.uni += dr * a.s_pack;
    a.u_ws += dr * a.s_eps_u;
    if (a.u_out) a.u_out += dr * a.s_eps_u;
    if (a.omega_out) a.omega_out += dr * a.s_rff_eps;
    if (a.phase_out) a.phase_out += dr * a.s_rff_u;
  }
  int blk = blockIdx.x;
  const int tid = threadIdx.x;
  __shared__ float sEll[256], sVar[16];  // Do*Di <= 256, Do <= 16 for every compiled specialisation
  for (int e = tid; e < Do * Di; e += 256) sEll[e] = softplus_lower(a.raw_ell[e]);
  if (tid < Do) sVar[tid] = softplus_lower(a.raw_var[tid]);
  __syncthreads();
  auto ELL = [&](int d, int i) { return sEll[d * Di + i]; };   // ell[d,i]
  auto VAR = [&](int d) { return sVar[d]; };
  // omega[i,s,d] = eps[i,s,d] / ell[d,i]   (kernels.py:112-124)
  auto OM = [&](int i, int s, int d) { return a.rff_eps[((size_t)i * S + s) * Do + d] / ELL(d, i); };
  if (blk < a.nb_rff) {
    const int e = blk * 256 + tid;
    const int SJ = cdiv(S, 64);
    if (e >= SJ * Do * 64) return;
    const int lane = e & 63, dd = (e >> 6) % Do, j = (e >> 6) / Do;
    const int s = j * 64 + lane;
    const bool ok = s < S;
    if (a.kernel == 0) {
      const int RQ = cdiv(Di + 2, 4);
      const size_t base = (size_t)(j * Do + dd) * RQ;
      for (int i = 0; i < Di; ++i) put_rec(a.pack, base, lane, i, ok ? OM(i, s, dd) * GP_INV2PI : 0.f);
      put_rec(a.pack, base, lane, Di, ok ? a.rff_u[s * Do + dd] : 0.f);
      put_rec(a.pack, base, lane, Di + 1, ok ? sqrtf(VAR(dd) / (float)S) * a.rff_w[s * Do + dd] : 0.f);
      for (int f = Di + 2; f < 4 * RQ; ++f) put_rec(a.pack, base, lane, f, 0.f);
    } else {
      const int D = Do, i = dd;
      const int RQ = cdiv(2 * D + 3, 4);
      const size_t base = (size_t)(j * D + i) * RQ;
      // theta_si = u[s,i] + sum_k x_k omega[k,s,i]
      for (int k = 0; k < D; ++k) put_rec(a.pack, base, lane, k, ok ? OM(k, s, i) * GP_INV2PI : 0.f);
      put_rec(a.pack, base, lane, D, ok ? a.rff_u[s * D + i] : 0.f);
      put_rec(a.pack, base, lane, D + 1, ok ? a.rff_w[s * D + i] : 0.f);
      put_rec(a.pack, base, lane, D + 2, ok ? a.rff_w[(S + s) * D + i] : 0.f);
      // B[s,i,jj] = norm_{s,jj} delta_{i,jj} - (sum_k omega[i,s,k] omega[jj,s,k]) / norm_{s,jj},
      // norm_{s,jj} = sqrt(sum_k' omega[k',s,jj]^2)   (kernels.py:327-336)
      for (int jj = 0; jj < D; ++jj) {
        float v = 0.f;
        if (ok) {
          float n2 = 0.f, g = 0.f;
          for (int k = 0; k < D; ++k) {
            float o = OM(k, s, jj);
            n2 = fmaf(o, o, n2);
            g = fmaf(OM(i, s, k), OM(jj, s, k), g);
          }
          float nrm = sqrtf(n2);
          v = (((i == jj) ? nrm : 0.f) - g / nrm) * sqrtf(VAR(jj) / (float)S);
        }
        put_rec(a.pack, base, lane, D + 3 + jj, v);
      }
      for (int f = 2 * D + 3; f < 4 * RQ; ++f) put_rec(a.pack, base, lane, f, 0.f);
    }
    return;
  }
  blk -= a.nb_rff;
  if (blk < a.nb_u) {
    // u[n,d] = sum_{m<=n} Us[d, n(n+1)/2 + m] eps_u[m,d] + Um[n,d]   (svpy.py:94-100, transforms.py:71-77)
    const int row = blk * 4 + (tid >> 6), lane = tid & 63;
    if (row >= M * Do) return;
    const int n = row / Do, d = row % Do;
    const float* us = a.Us + (size_t)d * ((size_t)M * (M + 1) / 2) + (size_t)n * (n + 1) / 2;
    float acc = 0.f;
    for (int m = lane; m <= n; m += 64) acc = fmaf(us[m], a.eps_u[m * Do + d], acc);
    acc = wave_allreduce_sum(acc) + a.Um[row];
    if (lane == 0) {
      a.u_ws[row] = acc;
      if (a.u_out) a.u_out[row] = acc;
    }
    return;
  }
  blk -= a.nb_u;
  if (blk < a.nb_ind) {
    const int e = blk * 256 + tid;
    const int RQ2 = cdiv(Di + Do, 4);
    if (e >= cdiv(M, 64) * 64) return;
    const int lane = e & 63, j = e >> 6;
    const int m = j * 64 + lane;
    const size_t base = (size_t)j * RQ2;
    for (int i = 0; i < Di; ++i) put_rec(a.pack_ind, base, lane, i, m < M ? a.Z[m * Di + i] : 0.f);
    for (int f = Di; f < 4 * RQ2; ++f) put_rec(a.pack_ind, base, lane, f, 0.f);
    return;
  }
  blk -= a.nb_ind;
  if (blk < a.nb_hyp) {
    // status word, then the smallest / largest diagonal entry of the factor (float bit patterns; positive floats order like
    // unsigned integers, so atomicMin / atomicMax on the bits work)
    if (a.info && threadIdx.x < 4) a.info[threadIdx.x] = threadIdx.x == 1 ? 0x7f800000 : 0;
    for (int e = tid; e < Do * Di; e += 256) {
      float l = sEll[e];
      if (a.ell_ws) a.ell_ws[e] = l;
      if (a.ell_out) a.ell_out[e] = l;
      float il2 = 1.f / (l * l);
      a.uni[e] = -0.5f * GP_LOG2E * il2;            // RBF wl[d][i]  /  DF wab[a][b]
      if (a.kernel == 1) a.uni[Do * Di + e] = il2;  // DF il2[a][b]
    }
    for (int d = tid; d < Do; d += 256) {
      float v = sVar[d];
      if (a.var_ws) a.var_ws[d] = v;
      if (a.var_out) a.var_out[d] = v;
      if (a.kernel == 1) a.uni[2 * Do * Di + d] = v;
    }
    return;
  }
  blk -= a.nb_hyp;
  {
    const int e = blk * 256 + tid;
    if (a.omega_out && e < Di * S * Do) {
      const int d = e % Do, i = e / (S * Do);
      a.omega_out[e] = a.rff_eps[e] / ELL(d, i);
    }
    if (a.phase_out && e < S * Do) a.phase_out[e] = (a.rff_u[e] * 2.f) * 3.14159265358979323846f;  // kernels.py:137
  }
}

// ---------------------------------------------------------------------------------------------
// K(Z) + jitter I, augmented with the rhs row (row n) and identity padding up to np.
//   A: (batch, np, np) row-major.  RBF: batch = Do, n = M.  DF: batch = 1, n = M D.
// ---------------------------------------------------------------------------------------------
__global__ void k_Kzz_rbf(int Di, int Do, int M, int np, const float* __restrict__ Z, const float* __restrict__ ell,
                          const float* __restrict__ var, const float* __restrict__ u_prior, float* __restrict__ A, int nd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y, d = blockIdx.z;
  if (c >= np) return;
  float v;
  if (r < M && c < M) {
    float q = 0.f;
    for (int i = 0; i < Di; ++i) {
      float t = (Z[r * Di + i] - Z[c * Di + i]) / ell[d * Di + i];
      q = fmaf(t, t, q);
    }
    v = var[d] * expf(-0.5f * q) + (r == c ? kJitter : 0.f);
  } else if (r < M + nd) {                           // rhs row of draw r - M (huge diagonal: eliminating its column touches nothing)
    v = c < M ? u_prior[(size_t)(r - M) * M * Do + c * Do + d] : (c == r ? 1e30f : 0.f);
  } else {
    v = (r == c) ? 1.f : 0.f;
  }
  A[((size_t)d * np + r) * np + c] = v;
}

__global__ void k_Kzz_df(int D, int M, int np, const float* __restrict__ Z, const float* __restrict__ ell,
                         const float* __restrict__ var, const float* __restrict__ u_prior, float* __restrict__ A, int nd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  const int n = M * D;
  if (c >= np) return;
  float v;
  if (r < n && c < n) {
    // row (nn,a), col (mm,b): K4[nn,mm,a,b], delta = z_mm - z_nn   (kernels.py:289-303)
    const int nn = r / D, a = r % D, mm = c / D, b = c % D;
    float r2 = 0.f;
    for (int i = 0; i < D; ++i) { float t = Z[mm * D + i] - Z[nn * D + i]; r2 = fmaf(t, t, r2); }
    float l = ell[a * D + b];
    float il2 = 1.f / (l * l);
    float da = Z[mm * D + a] - Z[nn * D + a], db = Z[mm * D + b] - Z[nn * D + b];
    float term = da * db * il2 + ((a == b) ? ((float)(D - 1) - r2 * il2) : 0.f);
    v = var[b] * expf(-0.5f * r2 * il2) * term * il2 + (r == c ? kJitter : 0.f);
  } else if (r < n + nd) {
    v = c < n ? u_prior[(size_t)(r - n) * n + c] : (c == r ? 1e30f : 0.f);
  } else {
    v = (r == c) ? 1.f : 0.f;
  }
  A[(size_t)r * np + c] = v;
}

// ---------------------------------------------------------------------------------------------
// Right-looking blocked Cholesky, NB = 32, one launch per block column k.
// grid.x enumerates the tiles (i,j), k <= j <= i, of the trailing matrix; grid.y = batch; block = 256.
// Every workgroup loads A_kk, A_ik, A_jk (trailing-updated so far), factors the diagonal tile and
// solves its two panel tiles in ONE LDS pivot loop (a 96 x 32 tall panel, one barrier per pivot, all
// 256 threads), then either publishes a result tile (j == k) or applies the rank-32 update to A_ij.
// The work per launch is constant (no K loop), so the chain is nblk x (one tile round trip).
// Results go to separate storage (Lmat for L_ik, Dfac for L_kk): the unfactored column-k tiles are
// still being read by the other workgroups of the launch, so nothing they read is written in place.
// ---------------------------------------------------------------------------------------------
#define GP_BCAST(v, l) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (l)))

// Cholesky of a 32x32 tile fused with the triangular solve of one 32x32 panel tile, in the registers of
// ONE wavefront: lanes 0..31 hold the rows of the diagonal tile D, lanes 32..63 the rows of the panel
// tile R.  Eliminating column j scales it by 1/sqrt(pivot) and subtracts row[j] * L[c][j] from every
// later column c -- the same instruction stream turns D into L (lower part) and R into R L^-T, so the
// panel solve costs no extra instructions.  Broadcasts are v_readlane (SGPR), nothing touches LDS.
// Broadcast columns C..C+3 of the pivot column and apply the rank-1 update to them.
// The v_readlane_b32 are pinned (inline asm) right in front of their FMAs: left to itself the scheduler
// hoists all 31 broadcasts of a pivot to the top, runs out of SGPRs and spills them through v_writelane.
// gfx950 needs wait states between a VALU SGPR write and a VALU read of that SGPR; hipcc does not see
// inside asm, so every group pads itself: in a full group the later v_readlane are the padding, short
// tail groups end in an explicit s_nop.
template <int J, int C> __device__ __forceinline__ void chol_cols_group(float (&row)[NB], float nsc) {
  if constexpr (C < NB) {
    constexpr int G = (NB - C) < 4 ? (NB - C) : 4;
    float l0 = 0.f, l1 = 0.f, l2 = 0.f, l3 = 0.f;
    // every group opens with s_nop 1: its input row[J] (first group) may have been written by the VALU
    // instruction right before, and a v_readlane of a just-written VGPR needs wait states that hipcc does
    // not insert in front of an asm statement (without it the broadcast returns the STALE register)
    if constexpr (G == 4) {
      asm volatile("s_nop 1\n\tv_readlane_b32 %0, %4, %5\n\tv_readlane_b32 %1, %4, %6\n\tv_readlane_b32 %2, %4, %7\n\tv_readlane_b32 %3, %4, %8"
                   : "=s"(l0), "=s"(l1), "=s"(l2), "=s"(l3) : "v"(row[J]), "n"(C), "n"(C + 1), "n"(C + 2), "n"(C + 3));
    } else if constexpr (G == 3) {
      asm volatile("s_nop 1\n\tv_readlane_b32 %0, %3, %4\n\tv_readlane_b32 %1, %3, %5\n\tv_readlane_b32 %2, %3, %6\n\ts_nop 1"
                   : "=s"(l0), "=s"(l1), "=s"(l2) : "v"(row[J]), "n"(C), "n"(C + 1), "n"(C + 2));
    } else if constexpr (G == 2) {
      asm volatile("s_nop 1\n\tv_readlane_b32 %0, %2, %3\n\tv_readlane_b32 %1, %2, %4\n\ts_nop 2"
                   : "=s"(l0), "=s"(l1) : "v"(row[J]), "n"(C), "n"(C + 1));
    } else {
      asm volatile("s_nop 1\n\tv_readlane_b32 %0, %1, %2\n\ts_nop 3" : "=s"(l0) : "v"(row[J]), "n"(C));
    }
    row[C] = fmaf(nsc, l0, row[C]);
    if constexpr (G > 1) row[C + 1] = fmaf(nsc, l1, row[C + 1]);
    if constexpr (G > 2) row[C + 2] = fmaf(nsc, l2, row[C + 2]);
    if constexpr (G > 3) row[C + 3] = fmaf(nsc, l3, row[C + 3]);
  }
}

template <int J, int C> __device__ __forceinline__ void chol_cols(float (&row)[NB], float nsc) {
  if constexpr (C < NB) {
    chol_cols_group<J, C>(row, nsc);
    chol_cols<J, C + 4>(row, nsc);
  }
}

// 1 / sqrt(piv) to full fp32 accuracy (v_rsq + one Newton step)
__device__ __forceinline__ float chol_rsqrt(float piv) {
  float inv = __builtin_amdgcn_rsqf(piv);
  return inv * (1.5f - 0.5f * piv * inv * inv);
}

// Pivot J.  Its scalar chain (broadcast of the diagonal, v_rsq, Newton step: ~8 dependent instructions, 60-80 cycles for a
// lone wavefront) was computed by pivot J-1 right after J-1's update of column J -- i.e. under J-1's remaining column updates,
// which do not depend on it -- and arrives as (piv, inv).  Same operations on the same values in the same order per element as
// the straight loop: results are bit-identical.
template <int J> __device__ __forceinline__ void chol_pivots(float (&row)[NB], int lane, bool& bad, float piv, float inv) {
  if constexpr (J < NB) {
    const float sc = row[J] * inv;
    row[J] = (lane == J) ? piv * inv : sc;
    const float nsc = -row[J];
    float piv_n = 1.f, inv_n = 1.f;
    if constexpr (J + 1 < NB) {
      chol_cols_group<J, J + 1>(row, nsc);           // columns J+1 .. J+4: column J+1 is final for pivot J+1
      piv_n = GP_BCAST(row[J + 1], J + 1);
      bad |= !(piv_n > 0.f);
      inv_n = chol_rsqrt(piv_n);
      chol_cols<J, J + 5>(row, nsc);                 // the other columns, independent of the chain above
    }
    chol_pivots<J + 1>(row, lane, bad, piv_n, inv_n);
  }
}

// The same elimination with the division in front (LDL^T order): the multiplier column of pivot J is w = a_J / p_J, the update
// a_C -= w * a_J[C] broadcasts the UNSCALED column (every broadcast of a pivot can issue as soon as the pivot starts), and the
// scaling of the finished column by rsqrt(p_J) -- its final Cholesky values, sqrt(p_J) on the diagonal since a_J[J] = p_J --
// leaves the dependent chain: per pivot it is  w = a_J * (1/p) -> update of column J+1 -> broadcast of p_(J+1) -> v_rcp + one
// Newton step  (6 instructions; the classic order above has 12: v_rsq, Newton, scale, select, negate in line).  The 32 pivots
// are a serial chain (tools/draw_probe.py: ~300 cycles per pivot, chain-bound from pivot ~12 on), so its length is the time.
__device__ __forceinline__ float chol_rcp(float p) {
  const float r = __builtin_amdgcn_rcpf(p);
  return r * fmaf(-p, r, 2.f);
}
template <int J> __device__ __forceinline__ void ldl_pivots(float (&row)[NB], bool& bad, float piv, float rinv) {
  if constexpr (J < NB) {
    const float nw = -(row[J] * rinv);
    float piv_n = 1.f, rinv_n = 1.f;
    if constexpr (J + 1 < NB) {
      chol_cols_group<J, J + 1>(row, nw);            // columns J+1 .. J+4 (broadcasts of the unscaled row[J])
      piv_n = GP_BCAST(row[J + 1], J + 1);
      bad |= !(piv_n > 0.f);
      rinv_n = chol_rcp(piv_n);
      chol_cols<J, J + 5>(row, nw);
    }
    row[J] *= chol_rsqrt(piv);                       // off the chain; lane J: p * rsqrt(p) = sqrt(p)
    ldl_pivots<J + 1>(row, bad, piv_n, rinv_n);
  }
}

// returns true when a pivot was not positive (matrix not positive definite)
__device__ __forceinline__ bool chol32_panel_wave(float (&row)[NB], int lane) {
  const float piv = GP_BCAST(row[0], 0);
  bool bad = !(piv > 0.f);
#ifdef GPODE_CHOL_CLASSIC
  chol_pivots<0>(row, lane, bad, piv, chol_rsqrt(piv));
#else
  ldl_pivots<0>(row, bad, piv, chol_rcp(piv));
#endif
  return bad;
}

__global__ __launch_bounds__(256) void k_chol_rl(float* __restrict__ Aall, float* __restrict__ Lall, int np,
                                                  size_t batch_stride, float* __restrict__ Dfac_all, size_t dfac_stride,
                                                  int k, int* __restrict__ info, int nblk, int jlim, int nreal) {
  __shared__ float sD[NB][NB + 1], sI[NB][NB + 1], sJ[NB][NB + 1];     // loaded tiles A_kk, A_ik, A_jk
  __shared__ float lD[NB][NB + 1], lI[NB][NB + 1], lJ[NB][NB + 1];     // factored: L_kk, L_ik, L_jk
  float* A = Aall + (size_t)blockIdx.y * batch_stride;
  float* Lm = Lall + (size_t)blockIdx.y * batch_stride;
  float* Dfac = Dfac_all + (size_t)blockIdx.y * dfac_stride + (size_t)k * NB * NB;
  // tile decode: t -> (ii, jj), jj <= ii, row-major over the lower triangle
  const int t = blockIdx.x;
  int i, j;
  if (jlim >= nblk) {
    int ii = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
    while (ii * (ii + 1) / 2 > t) --ii;
    i = k + ii; j = k + (t - ii * (ii + 1) / 2);
  } else {
    // panelled factorisation: only the tile columns k <= j < jlim of the current 128-wide panel are updated here,
    // column by column; everything to the right gets the whole panel's update from k_syrk_mfma afterwards
    int rem = t;
    j = k;
    while (rem >= nblk - j) { rem -= nblk - j; ++j; }
    i = j + rem;
  }
  const bool hasI = i > k, hasJ = (j > k) && (j != i);
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int c0 = k * NB;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q;
    sD[r][tx] = A[(size_t)(c0 + r) * np + c0 + tx];
    sI[r][tx] = hasI ? A[(size_t)(i * NB + r) * np + c0 + tx] : 0.f;
    sJ[r][tx] = hasJ ? A[(size_t)(j * NB + r) * np + c0 + tx] : 0.f;
  }
  __syncthreads();
  const int wv = tid >> 6, lane = tid & 63;
  if (wv == 0 || (wv == 1 && hasJ)) {
    // wave 0: [D ; A_ik], wave 1: [D ; A_jk] (D refactored redundantly, no cross-wave traffic)
    const int r = lane & 31;
    const bool panel = lane >= 32;
    float row[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) row[c] = panel ? (wv == 0 ? sI[r][c] : sJ[r][c]) : sD[r][c];
    const bool bad = chol32_panel_wave(row, lane);
    if (bad && wv == 0 && t == 0 && lane == 0) atomicOr(info, 1);  // not positive definite (reference raises)
    // branch-free write-back: panel halves go to lI / lJ, D halves to lD (both waves hold the same L_kk;
    // the duplicate store writes identical values)
    float* dst = panel ? (wv == 0 ? &lI[r][0] : &lJ[r][0]) : &lD[r][0];
#pragma unroll
    for (int c = 0; c < NB; ++c) dst[c] = (panel || c <= r) ? row[c] : 0.f;
  }
  __syncthreads();
  if (j == k) {
    // column-k tiles: publish the factor
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q;
      if (i == k) {
        Dfac[r * NB + tx] = lD[r][tx];
        if (r == tx && c0 + r < nreal) {             // diagonal of the factor proper (not the rhs row / identity padding)
          atomicMin(reinterpret_cast<unsigned*>(info) + 1, __float_as_uint(lD[r][tx]));
          atomicMax(reinterpret_cast<unsigned*>(info) + 2, __float_as_uint(lD[r][tx]));
        }
      } else Lm[(size_t)(i * NB + r) * np + c0 + tx] = lI[r][tx];
    }
  } else {
    // trailing tile: A_ij -= L_ik L_jk^T   (j == i: L_jk is L_ik)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q;
      float acc = 0.f;
      if (hasJ) {
#pragma unroll 8
        for (int p = 0; p < NB; ++p) acc = fmaf(lI[r][p], lJ[tx][p], acc);
      } else {
#pragma unroll 8
        for (int p = 0; p < NB; ++p) acc = fmaf(lI[r][p], lI[tx][p], acc);
      }
      float* dst = A + (size_t)(i * NB + r) * np + j * NB + tx;
      *dst = *dst - acc;
    }
  }
}

// TWO block columns (k, k+1) per launch -- the launch chain of a mid-sized factor (n = 600: 19 dependent launches of ~11 us) is
// bound by launch + factorisation latency, not by work, so a workgroup redoes what it would otherwise wait a launch for:
//   stage 1  [D_k ; A_ik], [D_k ; A_jk], [D_k ; A_(k+1)k] -> L_ik, L_jk, L_1        (three wavefronts side by side)
//   stage 2  E = A_(k+1)(k+1) - L_1 L_1^T,  Q_i = A_i(k+1) - L_ik L_1^T,  Q_j likewise   (the column k+1 after step k)
//   stage 3  [E ; Q_i], [E ; Q_j] -> L_(k+1)(k+1), L_i(k+1), L_j(k+1)
//   stage 4  A_ij -= L_ik L_jk^T + L_i(k+1) L_j(k+1)^T
// Tiles of column k stop after stage 1 and publish L_ik, tiles of column k+1 after stage 3 and publish L_i(k+1).  Nothing a
// workgroup reads is written in place by another one (columns k and k+1 of A are only read; factors go to Lmat / Dfac).
__device__ __forceinline__ void tile_load(float (&dst)[NB][NB + 1], const float* __restrict__ A, int np, int r0, int c0, int tx, int ty) {
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[ty + 8 * q][tx] = A[(size_t)(r0 + ty + 8 * q) * np + c0 + tx];
}
// dst = src - X Y^T (32 x 32 tiles in LDS), 256 threads
__device__ __forceinline__ void tile_sub_xyt(float (&dst)[NB][NB + 1], const float (&src)[NB][NB + 1], const float (&X)[NB][NB + 1],
                                             const float (&Y)[NB][NB + 1], int tx, int ty) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q;
    float acc = 0.f;
#pragma unroll 8
    for (int p = 0; p < NB; ++p) acc = fmaf(X[r][p], Y[tx][p], acc);
    dst[r][tx] = src[r][tx] - acc;
  }
}
// one wavefront: factor [Dt ; Pt] (rows of the diagonal tile in lanes 0..31, of the panel tile in lanes 32..63); the panel half goes
// to Lp, the diagonal half (lower triangle, zeros above) to Ld when given
__device__ __forceinline__ bool tile_factor(const float (&Dt)[NB][NB + 1], const float (&Pt)[NB][NB + 1], float (&Lp)[NB][NB + 1],
                                            float (*Ld)[NB + 1], int lane) {
  const int r = lane & 31;
  const bool panel = lane >= 32;
  float row[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) row[c] = panel ? Pt[r][c] : Dt[r][c];
  const bool bad = chol32_panel_wave(row, lane);
  if (panel) {
#pragma unroll
    for (int c = 0; c < NB; ++c) Lp[r][c] = row[c];
  } else if (Ld) {
#pragma unroll
    for (int c = 0; c < NB; ++c) Ld[r][c] = c <= r ? row[c] : 0.f;
  }
  return bad;
}

__global__ __launch_bounds__(256) void k_chol_rl2(float* __restrict__ Aall, float* __restrict__ Lall, int np, size_t batch_stride,
                                                   float* __restrict__ Dfac_all, size_t dfac_stride, int k, int* __restrict__ info,
                                                   int nblk, int nreal) {
  __shared__ float sD[NB][NB + 1], sPi[NB][NB + 1], sPj[NB][NB + 1], sP1[NB][NB + 1];      // column k: A_kk, A_ik, A_jk, A_(k+1)k
  __shared__ float sE[NB][NB + 1], sQi[NB][NB + 1], sQj[NB][NB + 1];                        // column k+1: A_(k+1)(k+1), A_i(k+1), A_j(k+1)
  __shared__ float lD[NB][NB + 1], lPi[NB][NB + 1], lPj[NB][NB + 1], lP1[NB][NB + 1];      // L_kk, L_ik, L_jk, L_(k+1)k
  __shared__ float lE[NB][NB + 1], lQi[NB][NB + 1], lQj[NB][NB + 1];                        // L_(k+1)(k+1), L_i(k+1), L_j(k+1)
  float* A = Aall + (size_t)blockIdx.y * batch_stride;
  float* Lm = Lall + (size_t)blockIdx.y * batch_stride;
  float* DfacB = Dfac_all + (size_t)blockIdx.y * dfac_stride;
  const int t = blockIdx.x;
  int ii = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
  while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
  while (ii * (ii + 1) / 2 > t) --ii;
  const int i = k + ii, j = k + (t - ii * (ii + 1) / 2);            // tile (i, j), k <= j <= i
  const int k1 = k + 1;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5, wv = tid >> 6, lane = tid & 63;
  const int c0 = k * NB, c1 = k1 * NB;
  const bool colk = j == k;                            // stops after stage 1
  const bool colk1 = j == k1;                          // stops after stage 3
  const bool needI = i > k, needJ = j > k && j != i;   // panel tiles of column k this workgroup factors
  const bool need1 = !colk && i != k1 && j != k1;      // L_(k+1)k is not already one of them
  tile_load(sD, A, np, c0, c0, tx, ty);
  if (needI) tile_load(sPi, A, np, i * NB, c0, tx, ty);
  if (needJ) tile_load(sPj, A, np, j * NB, c0, tx, ty);
  if (need1) tile_load(sP1, A, np, c1, c0, tx, ty);
  if (!colk) {
    tile_load(sE, A, np, c1, c1, tx, ty);
    if (i > k1) tile_load(sQi, A, np, i * NB, c1, tx, ty);
    if (j > k1 && j != i) tile_load(sQj, A, np, j * NB, c1, tx, ty);
  }
  __syncthreads();
  // ---- stage 1 ----
  bool bad = false;
  if (wv == 0) bad = tile_factor(sD, needI ? sPi : sD, lPi, lD, lane);     // (i == k: the panel half refactors D, unused)
  else if (wv == 1 && needJ) tile_factor(sD, sPj, lPj, nullptr, lane);
  else if (wv == 2 && need1) tile_factor(sD, sP1, lP1, nullptr, lane);
  if (bad && t == 0 && lane == 0 && wv == 0) atomicOr(info, 1);
  __syncthreads();
  if (colk) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q;
      if (i == k) {
        DfacB[(size_t)k * NB * NB + r * NB + tx] = lD[r][tx];
        if (r == tx && c0 + r < nreal) {
          atomicMin(reinterpret_cast<unsigned*>(info) + 1, __float_as_uint(lD[r][tx]));
          atomicMax(reinterpret_cast<unsigned*>(info) + 2, __float_as_uint(lD[r][tx]));
        }
      } else Lm[(size_t)(i * NB + r) * np + c0 + tx] = lPi[r][tx];
    }
    return;
  }
  // L_(k+1)k under its three names
  const float (&L1)[NB][NB + 1] = (i == k1) ? lPi : ((j == k1) ? lPj : lP1);
  // (j == k1 && j != i: needJ factored A_jk = A_(k+1)k into lPj;  i == k1 (then j == k1 too): it is lPi)
  // ---- stage 2: column k+1 after step k ----
  tile_sub_xyt(sE, sE, L1, L1, tx, ty);
  if (i > k1) tile_sub_xyt(sQi, sQi, lPi, L1, tx, ty);
  if (j > k1 && j != i) tile_sub_xyt(sQj, sQj, lPj, L1, tx, ty);
  __syncthreads();
  // ---- stage 3 ----
  bool bad1 = false;
  if (wv == 0) bad1 = tile_factor(sE, i > k1 ? sQi : sE, lQi, lE, lane);
  else if (wv == 1 && j > k1 && j != i) tile_factor(sE, sQj, lQj, nullptr, lane);
  if (bad1 && t == 2 && lane == 0 && wv == 0) atomicOr(info, 1);        // t == 2 is tile (k+1, k+1)
  __syncthreads();
  if (colk1) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q;
      if (i == k1) {
        DfacB[(size_t)k1 * NB * NB + r * NB + tx] = lE[r][tx];
        if (r == tx && c1 + r < nreal) {
          atomicMin(reinterpret_cast<unsigned*>(info) + 1, __float_as_uint(lE[r][tx]));
          atomicMax(reinterpret_cast<unsigned*>(info) + 2, __float_as_uint(lE[r][tx]));
        }
      } else Lm[(size_t)(i * NB + r) * np + c1 + tx] = lQi[r][tx];
    }
    return;
  }
  // ---- stage 4: trailing tile (j >= k + 2) ----
  const float (&Lj0)[NB][NB + 1] = (j == i) ? lPi : lPj;
  const float (&Lj1)[NB][NB + 1] = (j == i) ? lQi : lQj;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q;
    float acc = 0.f;
#pragma unroll 8
    for (int p = 0; p < NB; ++p) acc = fmaf(lPi[r][p], Lj0[tx][p], acc);
#pragma unroll 8
    for (int p = 0; p < NB; ++p) acc = fmaf(lQi[r][p], Lj1[tx][p], acc);
    float* dst = A + (size_t)(i * NB + r) * np + j * NB + tx;
    *dst = *dst - acc;
  }
}

// Trailing update of the panelled factorisation on the matrix cores: for 128 x 128 tiles (tm >= tn) of the rows / columns
// from r0 on, A -= P P^T with P = the 128 finished columns c0.. of L.  4 wavefronts x (4 x 4) v_mfma_f32_16x16x4_f32 tiles,
// both operands staged [k][128 + 16] (k rows of an operand fetch on disjoint banks), next k-tile prefetched to registers.
typedef float cf32x4 __attribute__((ext_vector_type(4)));
constexpr int ST = 128, SK = 16, SLD = ST + 16;

__global__ __launch_bounds__(256, 2) void k_syrk_mfma(float* __restrict__ Aall, const float* __restrict__ Lall, int np,
                                                       size_t batch_stride, int r0, int c0) {
  __shared__ __attribute__((aligned(16))) float sA[SK][SLD], sB[SK][SLD];
  float* A = Aall + (size_t)blockIdx.y * batch_stride;
  const float* Lm = Lall + (size_t)blockIdx.y * batch_stride;
  const int t = blockIdx.x;
  int tm = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
  while ((tm + 1) * (tm + 2) / 2 <= t) ++tm;
  while (tm * (tm + 1) / 2 > t) --tm;
  const int tn = t - tm * (tm + 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  cf32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = cf32x4{0.f, 0.f, 0.f, 0.f};
  const float* Pa = Lm + ((size_t)r0 + (size_t)tm * ST) * np + c0;
  const float* Pb = Lm + ((size_t)r0 + (size_t)tn * ST) * np + c0;
  float4 ra[2], rb[2];
  auto load = [&](int kt) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 256 * q, m = e >> 2, k4 = (e & 3) * 4;
      ra[q] = *reinterpret_cast<const float4*>(&Pa[(size_t)m * np + kt * SK + k4]);
      rb[q] = *reinterpret_cast<const float4*>(&Pb[(size_t)m * np + kt * SK + k4]);
    }
  };
  load(0);
  for (int kt = 0; kt < ST / SK; ++kt) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 256 * q, m = e >> 2, k4 = (e & 3) * 4;
      sA[k4 + 0][m] = ra[q].x; sA[k4 + 1][m] = ra[q].y; sA[k4 + 2][m] = ra[q].z; sA[k4 + 3][m] = ra[q].w;
      sB[k4 + 0][m] = rb[q].x; sB[k4 + 1][m] = rb[q].y; sB[k4 + 2][m] = rb[q].z; sB[k4 + 3][m] = rb[q].w;
    }
    __syncthreads();
    if (kt + 1 < ST / SK) load(kt + 1);
#pragma unroll
    for (int ks = 0; ks < SK / 4; ++ks) {
      float af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { af[i] = sA[4 * ks + lk][wm + 16 * i + lr]; bf[i] = sB[4 * ks + lk][wn + 16 * i + lr]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  float* C = A + ((size_t)r0 + (size_t)tm * ST) * np + r0 + (size_t)tn * ST;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* dst = C + (size_t)(wm + 16 * i + 4 * lk + r) * np + wn + 16 * j + lr;
        *dst = *dst - acc[i][j][r];
      }
}

// nu = L^-T (u - y),  y = row n of the factor (forward-solved rhs).  grid = batch, block = 256.
//   u element j of batch b at u[j * u_stride + b * u_bstride]; nu written dense (batch, n) and, scaled, into
//   the coefficient fields of the pack's inducing records (RBF: var_d nu[d,m]; DF: nu[(m,a)]).
// Right-looking block back-substitution.  CP4 > 0: every thread owns 4*CP4 consecutive columns and keeps
// the block row of L it needs NEXT in registers: the float4 loads for step k-1 are issued right after
// step k's update and land while wave 0 solves the next 32x32 diagonal system (v_readlane broadcasts), so
// global-load latency is off the dependency chain.  CP4 == 0: any n, loads issued inside the chain.
template <int CP4>
__global__ __launch_bounds__(256) void k_solve_back(const float* __restrict__ Aall, int n, int np, size_t batch_stride,
                                                     const float* __restrict__ Dfac_all, size_t dfac_stride,
                                                     const float* __restrict__ u, int u_stride, int u_bstride,
                                                     float* __restrict__ nu, float* __restrict__ nu_out,
                                                     int kernel, int Di, int Do, int M, const float* __restrict__ var,
                                                     float* __restrict__ pack_ind, size_t u_dstride, size_t nu_dstride,
                                                     size_t pack_dstride) {
  extern __shared__ __attribute__((aligned(16))) float sv[];  // np floats: residual, overwritten by the solution
  __shared__ float sx[NB];
  const int b = blockIdx.x, dr = blockIdx.y;          // system, Monte-Carlo draw (its rhs is row n + dr of the factor)
  const float* A = Aall + (size_t)b * batch_stride;
  const float* Dfac = Dfac_all + (size_t)b * dfac_stride;
  u += (size_t)dr * u_dstride;
  nu += (size_t)dr * nu_dstride;
  if (nu_out) nu_out += (size_t)dr * nu_dstride;
  pack_ind += (size_t)dr * pack_dstride;
  const int tid = threadIdx.x, lane = tid & 63;
  const int rr = n + dr, kl = rr / NB, cl = kl * NB;  // block holding the rhs row
  for (int j = tid; j < np; j += 256) {
    float y = 0.f;
    if (j < n) y = (j < cl) ? A[(size_t)rr * np + j] : Dfac[(size_t)kl * NB * NB + (rr - cl) * NB + (j - cl)];
    sv[j] = j < n ? u[(size_t)j * u_stride + (size_t)b * u_bstride] - y : 0.f;
  }
  const int nblk = cdiv(n, NB);
  constexpr int CP = CP4 > 0 ? CP4 : 1;
  float4 cur[CP][NB];
  auto prefetch = [&](int k) {  // rows of block k, this thread's columns (left of the diagonal block only)
    if (CP4 == 0 || k < 0) return;
    const int c0 = k * NB;
#pragma unroll
    for (int q = 0; q < CP; ++q) {
      const int c = 4 * (tid + 256 * q);
      if (c < c0) {
#pragma unroll
        for (int r = 0; r < NB; ++r) cur[q][r] = *reinterpret_cast<const float4*>(A + (size_t)(c0 + r) * np + c);
      }
    }
  };
  // lane c of wave 0 holds column c of the current diagonal block L_kk (fetched one step ahead)
  const int cc = lane & 31;
  float Lc[NB], Lc_nxt[NB];
  auto prefetch_diag = [&](float (&dst)[NB], int k) {
    if (k < 0 || tid >= 64) return;
    const float* Lk = Dfac + (size_t)k * NB * NB;
#pragma unroll
    for (int r = 0; r < NB; ++r) dst[r] = Lk[r * NB + cc];
  };
  prefetch(nblk - 1);
  prefetch_diag(Lc, nblk - 1);
  __syncthreads();
  for (int k = nblk - 1; k >= 0; --k) {
    const int c0 = k * NB;
    if (tid < 64) {
      prefetch_diag(Lc_nxt, k - 1);
      // 32x32 transposed solve in wave 0: lane c holds column c of L_kk and residual c
      const int c = cc;
      float res = sv[c0 + c];
      float d = 1.f;
#pragma unroll
      for (int r = 0; r < NB; ++r) d = (r == c) ? Lc[r] : d;
      const float myinv = 1.f / d;
#pragma unroll
      for (int r = NB - 1; r >= 0; --r) {
        // x_r = res_r / L_rr  (rows past n are padding: x = 0)
        float xr = GP_BCAST(res, r) * GP_BCAST(myinv, r);
        xr = (c0 + r < n) ? xr : 0.f;
        res = (c == r) ? xr : fmaf(-Lc[r], xr, res);  // lanes c < r consume L[r][c]; lanes c > r hold x already
      }
      if (lane < NB) { sx[lane] = res; sv[c0 + lane] = res; }
#pragma unroll
      for (int r = 0; r < NB; ++r) Lc[r] = Lc_nxt[r];
    }
    __syncthreads();
    if (CP4 > 0) {
#pragma unroll
      for (int q = 0; q < CP; ++q) {
        const int c = 4 * (tid + 256 * q);
        if (c < c0) {  // c0 is a multiple of 32: the four columns are all left of the diagonal block
          float4 acc = *reinterpret_cast<float4*>(sv + c);
#pragma unroll
          for (int r = 0; r < NB; ++r) {
            const float x = sx[r];
            acc.x = fmaf(-cur[q][r].x, x, acc.x); acc.y = fmaf(-cur[q][r].y, x, acc.y);
            acc.z = fmaf(-cur[q][r].z, x, acc.z); acc.w = fmaf(-cur[q][r].w, x, acc.w);
          }
          *reinterpret_cast<float4*>(sv + c) = acc;
        }
      }
      prefetch(k - 1);
    } else {
      for (int c = tid; c < c0; c += 256) {
        float acc = sv[c];
#pragma unroll 8
        for (int r = 0; r < NB; ++r) acc = fmaf(-A[(size_t)(c0 + r) * np + c], sx[r], acc);
        sv[c] = acc;
      }
    }
    __syncthreads();
  }
  for (int j = tid; j < n; j += 256) {
    const float v = sv[j];
    nu[(size_t)b * n + j] = v;
    if (nu_out) nu_out[(size_t)b * n + j] = v;
    if (!pack_ind) continue;                         // kern.compute_nu on its own: no pack to publish into
    // coefficient field of the inducing record (gp_eval.hpp)
    const int RQ2 = cdiv(Di + Do, 4);
    int m, d;
    float coef;
    if (kernel == 0) { m = j; d = b; coef = var[d] * v; } else { m = j / Do; d = j % Do; coef = v; }
    const int field = Di + d;
    pack_ind[(((size_t)(m >> 6) * RQ2 + (field >> 2)) * 64 + (m & 63)) * 4 + (field & 3)] = coef;
  }
}

// k_solve_back for n <= 1024 with the loads off the chain for good: 1024 threads, thread c owns COLUMN c of the residual and keeps the
// block rows of the next THREE steps in registers (a load issued at the end of step k is used in step k - 3: ~2 us later, the latency
// of a line another XCD has just written), and all diagonal blocks sit in LDS from the start.  k_solve_back<1> fetched one step
// ahead with a quarter of its threads (4 columns each): every one of the 19 steps of the 600 x 600 DF system waited for its block
// row -- 57 us for 0.7 MB.  Same arithmetic in the same order per column, so the results are bit-identical to k_solve_back's.
constexpr int SBD_THREADS = 1024, SBD_STAGES = 3;
__global__ __launch_bounds__(SBD_THREADS) void k_solve_back_deep(const float* __restrict__ Aall, int n, int np, size_t batch_stride,
                                                                const float* __restrict__ Dfac_all, size_t dfac_stride,
                                                                const float* __restrict__ u, int u_stride, int u_bstride,
                                                                float* __restrict__ nu, float* __restrict__ nu_out,
                                                                int kernel, int Di, int Do, int M, const float* __restrict__ var,
                                                                float* __restrict__ pack_ind, size_t u_dstride, size_t nu_dstride,
                                                                size_t pack_dstride) {
  extern __shared__ __attribute__((aligned(16))) float sv[];  // np floats: residual, overwritten by the solution; then nblk diagonal blocks
  __shared__ __attribute__((aligned(16))) float sx[NB];
  const int b = blockIdx.x, dr = blockIdx.y;
  const float* A = Aall + (size_t)b * batch_stride;
  const float* Dfac = Dfac_all + (size_t)b * dfac_stride;
  u += (size_t)dr * u_dstride;
  nu += (size_t)dr * nu_dstride;
  if (nu_out) nu_out += (size_t)dr * nu_dstride;
  if (pack_ind) pack_ind += (size_t)dr * pack_dstride;
  const int tid = threadIdx.x, lane = tid & 63;
  const int nblk = cdiv(n, NB);
  float* sD = sv + np;
  float cur[SBD_STAGES][NB];
  auto prefetch = [&](float (&dst)[NB], int k) {      // rows of block k, this thread's column (left of the diagonal block only)
    if (k < 1 || tid >= k * NB) return;
    const float* src = A + (size_t)k * NB * np + tid;
#pragma unroll
    for (int r = 0; r < NB; ++r) dst[r] = src[(size_t)r * np];
  };
  prefetch(cur[0], nblk - 1);
  prefetch(cur[1], nblk - 2);
  prefetch(cur[2], nblk - 3);
  const int rr = n + dr, kl = rr / NB, cl = kl * NB;  // block holding the rhs row
  for (int j = tid; j < np; j += SBD_THREADS) {
    float y = 0.f;
    if (j < n) y = (j < cl) ? A[(size_t)rr * np + j] : Dfac[(size_t)kl * NB * NB + (rr - cl) * NB + (j - cl)];
    sv[j] = j < n ? u[(size_t)j * u_stride + (size_t)b * u_bstride] - y : 0.f;
  }
  for (int e = tid; e < nblk * NB * NB / 4; e += SBD_THREADS)
    reinterpret_cast<float4*>(sD)[e] = reinterpret_cast<const float4*>(Dfac)[e];
  __syncthreads();
  const int cc = lane & 31;
  auto step = [&](int k, float (&mine)[NB]) {
    const int c0 = k * NB;
    if (tid < 64) {
      // 32x32 transposed solve in wave 0: lane c holds residual c and reads column c of L_kk from LDS, eight rows at a time
      const float* Lk = sD + (size_t)k * NB * NB + cc;
      float res = sv[c0 + cc];
      const float myinv = 1.f / Lk[cc * NB];
#pragma unroll
      for (int rb = NB - 8; rb >= 0; rb -= 8) {
        float Lc[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) Lc[r] = Lk[(rb + r) * NB];
#pragma unroll
        for (int r8 = 7; r8 >= 0; --r8) {
          const int r = rb + r8;
          float xr = GP_BCAST(res, r) * GP_BCAST(myinv, r);
          xr = (c0 + r < n) ? xr : 0.f;
          res = (cc == r) ? xr : fmaf(-Lc[r8], xr, res);
        }
      }
      if (lane < NB) { sx[lane] = res; sv[c0 + lane] = res; }
    }
    __syncthreads();
    if (tid < c0) {
      float acc = sv[tid];
#pragma unroll
      for (int r4 = 0; r4 < NB / 4; ++r4) {
        const float4 x = reinterpret_cast<const float4*>(sx)[r4];
        acc = fmaf(-mine[4 * r4 + 0], x.x, acc); acc = fmaf(-mine[4 * r4 + 1], x.y, acc);
        acc = fmaf(-mine[4 * r4 + 2], x.z, acc); acc = fmaf(-mine[4 * r4 + 3], x.w, acc);
      }
      sv[tid] = acc;
    }
    prefetch(mine, k - SBD_STAGES);
    __syncthreads();
  };
  for (int k = nblk - 1; k >= 0; k -= SBD_STAGES) {
    step(k, cur[0]);
    if (k - 1 >= 0) step(k - 1, cur[1]);
    if (k - 2 >= 0) step(k - 2, cur[2]);
  }
  for (int j = tid; j < n; j += SBD_THREADS) {
    const float v = sv[j];
    nu[(size_t)b * n + j] = v;
    if (nu_out) nu_out[(size_t)b * n + j] = v;
    if (!pack_ind) continue;
    const int RQ2 = cdiv(Di + Do, 4);
    int m, d;
    float coef;
    if (kernel == 0) { m = j; d = b; coef = var[d] * v; } else { m = j / Do; d = j % Do; coef = v; }
    const int field = Di + d;
    pack_ind[(((size_t)(m >> 6) * RQ2 + (field >> 2)) * 64 + (m & 63)) * 4 + (field & 3)] = coef;
  }
}
inline bool solve_back_deep_fits(int np, int nblk) {
  static const bool off = [] { const char* e = getenv("GPODE_SOLVE_BACK_DEEP"); return e && e[0] == '0'; }();
  return !off && np <= SBD_THREADS && sizeof(float) * ((size_t)np + (size_t)nblk * NB * NB) <= 150 * 1024;
}

// ---------------------------------------------------------------------------------------------
// k_draw_lds: K(Z) + jitter I, its Cholesky factor and nu = L^-T (u - L^-1 f_prior(Z)) in ONE launch, one 512-thread
// workgroup per system (RBF: one per output dimension; DF: the single (M D)^2 system), the whole matrix resident in LDS
// (np <= 192: M <= 191 for RBF -- BASELINE configs[0], [2], [3]: six 128 x 128 systems of 66 KB).  Replaces the chain
// k_Kzz + nblk x k_chol_rl + k_solve_back (6 dependent launches at M = 100, ~11 us each) by their sum of arithmetic (~20 us):
//   fill      every thread evaluates kernel entries straight into LDS (tile rows >= tile columns), the rhs row n = f_prior(Z),
//             identity padding -- the formulas of k_Kzz_rbf / k_Kzz_df, term for term
//   factor    per block column k: wavefront w runs chol32_panel_wave on [D_k ; A_(k+1+w),k] (the register-resident fused
//             factor + panel solve of k_chol_rl), results go back to LDS in place; then the trailing tiles A_ij -= L_ik L_jk^T
//             on the matrix cores (v_mfma_f32_16x16x4_f32, operands read from LDS with row stride np + 2: the four 16-lane
//             groups of an operand fetch hit disjoint banks), 16 x 16 quadrants dealt to the 8 wavefronts
//   solve     block back-substitution in LDS (the wave-0 triangular solve of k_solve_back)
//   publish   factor tiles to Lmat / Dfac (the backward's L^-1 and the optional Lu read them), nu to ws / output / pack
// ---------------------------------------------------------------------------------------------
// phase timing (tools/draw_probe.py builds the library with -DDRAW_PROBE; compiled out otherwise)
#ifdef DRAW_PROBE
#define DPROBE(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) { const long long t_ = clock64(); dprobe[i] += t_ - dlast; dlast = t_; } } while (0)
#else
#define DPROBE(i)
#endif

template <int KERNEL>
__global__ __launch_bounds__(512) void k_draw_lds(int Di, int Do, int M, int n, int np, int nblk, const float* __restrict__ Z,
                                                   const float* __restrict__ ell, const float* __restrict__ var,
                                                   const float* __restrict__ u_prior, const float* __restrict__ u,
                                                   float* __restrict__ Lall, size_t batch_stride, float* __restrict__ Dfac_all,
                                                   size_t dfac_stride, float* __restrict__ nu_ws, float* __restrict__ nu_out,
                                                   float* __restrict__ pack_ind, int* __restrict__ info, int nd, size_t u_dstride,
                                                   size_t nu_dstride, size_t pack_dstride) {
  // nd Monte-Carlo draws share the factor: their right-hand sides f_prior_l(Z) are rows n .. n + nd - 1, u / nu / pack are per draw
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int LD = np + 2;
  float* sA = dsm;                                   // [np][LD]
  float* sv = dsm + (size_t)np * LD;                 // [nd][np] solutions
  float* sx = sv + (size_t)nd * np;                  // [NB]
  float* sZ = sx + NB;                               // [M][Di]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef DRAW_PROBE
  long long dprobe[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast = clock64();
#endif
  // Z transposed ([i][m], row stride M | 1: the per-lane column reads below are conflict-free) and the per-system constants
  const int ZS = M | 1;
  for (int e = tid; e < M * Di; e += 512) sZ[(e % Di) * ZS + e / Di] = Z[e];
  float* sC = sZ + (size_t)Di * ZS;                  // RBF: 1 / ell[b][i]; DF: 1 / ell[a][bb]^2 then var[bb]
  if (KERNEL == 0) {
    for (int e = tid; e < Di; e += 512) sC[e] = 1.f / ell[b * Di + e];
  } else {
    for (int e = tid; e < Do * Do; e += 512) { const float l = ell[e]; sC[e] = 1.f / (l * l); }
    for (int e = tid; e < Do; e += 512) sC[Do * Do + e] = var[e];
  }
  __syncthreads();
  // ---- fill: wavefront w takes rows w, w + 8, ...; its lanes the columns lane, lane + 64, ... of the tiles at or below the
  // row's diagonal tile.  The row's inducing point is wave-uniform (LDS broadcast), the column's is read once per chunk.
  // The coordinate loop is unrolled for the compiled latent widths (a runtime trip count serialises one LDS round trip per
  // coordinate: 19 of the kernel's first 55 us).
  auto fill = [&](auto dtag) {
    constexpr int DC = decltype(dtag)::value;        // compile-time Di, or 0: runtime
    constexpr int DA = DC > 0 ? DC : 1;
    const int DIr = DC > 0 ? DC : Di;
    const float vb = KERNEL == 0 ? var[b] : 0.f;
    // the lane's columns are the same for every row: their inducing points (and, RBF, the inverse lengthscales) stay in
    // registers when the width is a compile-time constant; only the row's point is fetched per row (LDS broadcast)
    float zc[3][DA], ie[DA], vq[3];
    int mq[3], bq[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int c = lane + 64 * q;
      mq[q] = KERNEL == 0 ? c : c / Do;
      bq[q] = KERNEL == 0 ? 0 : c - mq[q] * Do;
      const bool ok = c < n;
      if (!ok) { mq[q] = 0; bq[q] = 0; }
      vq[q] = KERNEL == 0 ? vb : sC[Do * Do + bq[q]];
      if (DC > 0) {
#pragma unroll
        for (int i = 0; i < DA; ++i) zc[q][i] = sZ[i * ZS + mq[q]];
      }
    }
    if (DC > 0 && KERNEL == 0) {
#pragma unroll
      for (int i = 0; i < DA; ++i) ie[i] = sC[i];
    }
    for (int r = wave; r < np; r += 8) {
      const int cend = ((r >> 5) + 1) * NB;          // first column right of the diagonal tile
      const int nn = KERNEL == 0 ? (r < n ? r : 0) : (r < n ? r / Do : 0), a = KERNEL == 0 ? 0 : (r < n ? r - nn * Do : 0);
      float zr[DA];
      if (DC > 0) {
#pragma unroll
        for (int i = 0; i < DA; ++i) zr[i] = sZ[i * ZS + nn];
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int c = lane + 64 * q;
        if (c >= cend) continue;
        float v;
        if (r < n && c < n) {
          if (KERNEL == 0) {
            float qq = 0.f;
            if (DC > 0) {
#pragma unroll
              for (int i = 0; i < DA; ++i) { const float t = (zr[i] - zc[q][i]) * ie[i]; qq = fmaf(t, t, qq); }
            } else {
              for (int i = 0; i < DIr; ++i) { const float t = (sZ[i * ZS + r] - sZ[i * ZS + c]) * sC[i]; qq = fmaf(t, t, qq); }
            }
            v = vb * expf(-0.5f * qq) + (r == c ? kJitter : 0.f);
          } else {
            const int D = DIr, mm = mq[q], bb = bq[q];
            float r2 = 0.f;
            if (DC > 0) {
#pragma unroll
              for (int i = 0; i < DA; ++i) { const float t = zc[q][i] - zr[i]; r2 = fmaf(t, t, r2); }
            } else {
              for (int i = 0; i < D; ++i) { const float t = sZ[i * ZS + mm] - sZ[i * ZS + nn]; r2 = fmaf(t, t, r2); }
            }
            const float il2 = sC[a * D + bb];
            const float da = sZ[a * ZS + mm] - sZ[a * ZS + nn], db = sZ[bb * ZS + mm] - sZ[bb * ZS + nn];
            const float term = da * db * il2 + ((a == bb) ? ((float)(D - 1) - r2 * il2) : 0.f);
            v = vq[q] * expf(-0.5f * r2 * il2) * term * il2 + (r == c ? kJitter : 0.f);
          }
        } else if (r < n + nd) {
          const float* up = u_prior + (size_t)(r - n) * u_dstride;
          v = c < n ? (KERNEL == 0 ? up[c * Do + b] : up[c]) : (c == r ? 1e30f : 0.f);
        } else {
          v = (r == c) ? 1.f : 0.f;
        }
        sA[r * LD + c] = v;
      }
    }
  };
  switch (Di) {
    case 2: fill(std::integral_constant<int, 2>{}); break;
    case 3: fill(std::integral_constant<int, 3>{}); break;
    case 4: fill(std::integral_constant<int, 4>{}); break;
    case 6: fill(std::integral_constant<int, 6>{}); break;
    case 8: fill(std::integral_constant<int, 8>{}); break;
    case 12: fill(std::integral_constant<int, 12>{}); break;
    case 16: fill(std::integral_constant<int, 16>{}); break;
    default: fill(std::integral_constant<int, 0>{}); break;
  }
  __syncthreads();
  DPROBE(0);
  // ---- factor ---------------------------------------------------------------------------------
  const int lr = lane & 15, lk = lane >> 4;
  for (int k = 0; k < nblk; ++k) {
    const int c0 = k * NB, below = nblk - k - 1;     // panel tiles under the diagonal tile
    // panel tiles are dealt round-robin to the wavefronts (np <= 192: at most 5, one round); wave 0 always runs (it owns D_k)
    for (int t0 = 0; t0 < (below > 0 ? below : 1); t0 += 8) {
      const int t = t0 + wave;
      const bool active = t < below || (wave == 0 && t0 == 0);
      float row[NB];
      const bool panel = lane >= 32;
      const int rr = lane & 31;
      const int grow = panel ? (t < below ? (k + 1 + t) * NB + rr : c0 + rr) : c0 + rr;
      if (active) {
#pragma unroll
        for (int c = 0; c < NB; ++c) row[c] = sA[grow * LD + c0 + c];
      }
      __syncthreads();                               // every wavefront holds its copy of D_k before anyone overwrites it
      if (active) {
        const bool bad = chol32_panel_wave(row, lane);
        if (bad && wave == 0 && t0 == 0 && lane == 0) atomicOr(info, 1);
        if (!panel) {
          if (wave == 0 && t0 == 0) {
#pragma unroll
            for (int c = 0; c < NB; ++c) sA[grow * LD + c0 + c] = c <= rr ? row[c] : 0.f;
          }
        } else if (t < below) {
#pragma unroll
          for (int c = 0; c < NB; ++c) sA[grow * LD + c0 + c] = row[c];
        }
      }
    }
    __syncthreads();
    DPROBE(1);
    // trailing update: tiles (i, j), k < j <= i < nblk, as 16 x 16 quadrants; job = (tile, quadrant)
    const int T = below * (below + 1) / 2;
    for (int job = wave; job < 4 * T; job += 8) {
      const int tt = job >> 2, qd = job & 3;
      int ii = (int)((sqrtf(8.f * (float)tt + 1.f) - 1.f) * 0.5f);
      while ((ii + 1) * (ii + 2) / 2 <= tt) ++ii;
      while (ii * (ii + 1) / 2 > tt) --ii;
      const int i = k + 1 + ii, j = k + 1 + (tt - ii * (ii + 1) / 2);
      const int r0 = i * NB + 16 * (qd >> 1), q0 = j * NB + 16 * (qd & 1);
      cf32x4 acc = cf32x4{0.f, 0.f, 0.f, 0.f};
      const float* pa = sA + (r0 + lr) * LD + c0 + lk;
      const float* pb = sA + (q0 + lr) * LD + c0 + lk;
#pragma unroll
      for (int ks = 0; ks < NB / 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * ks], pb[4 * ks], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float* dst = sA + (r0 + 4 * lk + r) * LD + q0 + lr;
        *dst = *dst - acc[r];
      }
    }
    __syncthreads();
    DPROBE(2);
  }
  for (int j = tid; j < n; j += 512) {               // diagonal range of the factor (conditioning estimate, gpode_cache_pivots)
    const float dj = sA[j * LD + j];
    atomicMin(reinterpret_cast<unsigned*>(info) + 1, __float_as_uint(dj));
    atomicMax(reinterpret_cast<unsigned*>(info) + 2, __float_as_uint(dj));
  }
  // ---- nu = L^-T (u - y), y = row n of the factor: block back-substitution by wavefront 0 alone -- the residual lives in its
  // registers (lane l holds entries l, l + 64, l + 128), the 32 solved unknowns of a block travel as wave-uniform values
  // (v_readlane), so the 2 x nblk workgroup barriers of a shared-residual formulation disappear from the chain.
  // One wavefront per draw (draws wave, wave + 8, ...): the substitutions of the draws are independent and run side by side.
  for (int dr = wave; dr < nd; dr += 8) {
    const float* ud = u + (size_t)dr * u_dstride;
    float res[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int j = lane + 64 * q;
      res[q] = j < n ? (KERNEL == 0 ? ud[(size_t)j * Do + b] : ud[j]) - sA[(n + dr) * LD + j] : 0.f;
    }
    for (int k = cdiv(n, NB) - 1; k >= 0; --k) {
      const int c0 = k * NB, q0 = c0 >> 6, l0 = c0 & 63;     // the block sits in chunk q0, lanes l0 .. l0 + 31
      const int c = (lane - l0) & 31;                        // column within the block (meaningful in the block's lanes)
      const bool mine = lane >= l0 && lane < l0 + NB;
      float Lc[NB];                                          // column c of L_kk
#pragma unroll
      for (int r = 0; r < NB; ++r) Lc[r] = sA[(c0 + r) * LD + c0 + c];
      float dg = 1.f;
#pragma unroll
      for (int r = 0; r < NB; ++r) dg = (r == c) ? Lc[r] : dg;
      // rows past n are padding (row n is the rhs row): their unknowns are 0, which a zero reciprocal gives for free
      const float myinv = (c0 + c < n) ? 1.f / dg : 0.f;
      // Scaled by the reciprocal diagonal up front, y_c = res_c / L_cc and Lp[r] = L[r][c] / L_cc, a row of the substitution is
      // broadcast -> one FMA: x_r = y_r when its turn comes, then y_c -= Lp[r] x_r for the columns left of it (the entries right
      // of the diagonal are stored as zeros, the lane of row r itself is done).  No multiply, select or mask on the 32-long chain.
#pragma unroll
      for (int r = 0; r < NB; ++r) Lc[r] *= myinv;
      float y = (q0 == 0 ? res[0] : (q0 == 1 ? res[1] : res[2])) * myinv;
      float xs[NB], rb = 0.f;
#pragma unroll
      for (int r = NB - 1; r >= 0; --r) {
        const float xr = GP_BCAST(y, l0 + r);
        xs[r] = xr;
        rb = (c == r) ? xr : rb;                             // this lane's own unknown (off the chain)
        y = fmaf(-Lc[r], xr, y);
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int j = lane + 64 * q;
        float v = res[q];
        if (q == q0 && mine) v = rb;                         // the solved block
        if (j < c0) {                                        // columns left of the block: subtract its contribution
#pragma unroll
          for (int r = 0; r < NB; ++r) v = fmaf(-sA[(c0 + r) * LD + j], xs[r], v);
        }
        res[q] = v;
      }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (lane + 64 * q < np) sv[(size_t)dr * np + lane + 64 * q] = res[q];
  }
  __syncthreads();
  DPROBE(3);
  // ---- publish --------------------------------------------------------------------------------
  float* Lm = Lall + (size_t)b * batch_stride;
  float* Dfac = Dfac_all + (size_t)b * dfac_stride;
  for (int r = wave; r < np; r += 8) {               // a wavefront writes runs of 64 consecutive floats of one row
    const int tr = r >> 5, cend = (tr + 1) * NB;
    for (int c = lane; c < cend; c += 64) {
      const float v = sA[r * LD + c];
      if ((c >> 5) < tr) Lm[(size_t)r * np + c] = v;
      else Dfac[(size_t)tr * NB * NB + (r & 31) * NB + (c & 31)] = v;
    }
  }
  for (int e = tid; e < nd * n; e += 512) {
    const int dr = e / n, j = e - dr * n;
    const float v = sv[(size_t)dr * np + j];
    nu_ws[(size_t)dr * nu_dstride + (size_t)b * n + j] = v;
    if (nu_out) nu_out[(size_t)dr * nu_dstride + (size_t)b * n + j] = v;
    const int RQ2 = cdiv(Di + Do, 4);
    int m, d;
    float coef;
    if (KERNEL == 0) { m = j; d = b; coef = var[d] * v; } else { m = j / Do; d = j % Do; coef = v; }
    const int field = Di + d;
    pack_ind[(size_t)dr * pack_dstride + (((size_t)(m >> 6) * RQ2 + (field >> 2)) * 64 + (m & 63)) * 4 + (field & 3)] = coef;
  }
#ifdef DRAW_PROBE
  __syncthreads();
  DPROBE(4);
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("k_draw_lds cycles: fill %lld  factor+solve-panel %lld  trailing %lld  back-subst %lld  publish %lld\n", dprobe[0], dprobe[1],
           dprobe[2], dprobe[3], dprobe[4]);
#endif
}

// LDS bytes of k_draw_lds; it takes systems up to np = 192 (GPODE_DRAW_CHAIN=1 keeps the launch chain for every size: A/B switch)
static inline size_t draw_lds_bytes(int np, int M, int Di, int nd) {   // sA, sv, sx, Z transposed, constants (<= Di * Di + Di)
  return sizeof(float) * ((size_t)np * (np + 2) + (size_t)nd * np + NB + (size_t)Di * (M | 1) + (size_t)Di * Di + Di);
}
static inline bool draw_in_lds(int np, int M, int Di, int nd) {
  static const bool off = [] { const char* e = getenv("GPODE_DRAW_CHAIN"); return e && e[0] == '1'; }();
  return !off && np <= 192 && draw_lds_bytes(np, M, Di, nd) <= 160 * 1024;
}

// ---------------------------------------------------------------------------------------------
// Back-substitution for big factors (the single-workgroup kernel above is bandwidth-starved at n = 8192: 134 MB of L through
// one CU).  One launch per 128-row panel, last panel first; workgroup c (<= P) owns residual entries 128 c .. 128 c + 127:
//   launch P:  y_c -= L[panel P+1][cols c]^T nu[panel P+1]   (every workgroup; a 128 x 128 GEMV, coalesced along the columns)
//              workgroup c == P then solves its 128 x 128 transposed triangular system and publishes nu[panel P].
// The first launch initialises y = u - (row n of the factor).  y lives in the first np floats of the (consumed) A buffer.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float factor_elem(const float* __restrict__ Lm, const float* __restrict__ Dfac, int np, int r, int c) {
  return (r / NB == c / NB) ? Dfac[(size_t)(r / NB) * NB * NB + (r % NB) * NB + c % NB] : Lm[(size_t)r * np + c];
}

__global__ __launch_bounds__(256) void k_solve_back_panel(const float* __restrict__ Lall, int n, int np, size_t batch_stride,
                                                           const float* __restrict__ Dfac_all, size_t dfac_stride,
                                                           const float* __restrict__ u, int u_stride, int u_bstride,
                                                           float* __restrict__ yall, float* __restrict__ nu, int P, int first,
                                                           size_t u_dstride, size_t nu_dstride) {
  constexpr int PW = 128;
  __shared__ float sx[PW], spart[PW];
  const int c = blockIdx.x, b = blockIdx.y, dr = blockIdx.z, tid = threadIdx.x;   // dr: Monte-Carlo draw, rhs = row n + dr of the factor
  const float* Lm = Lall + (size_t)b * batch_stride;
  const float* Dfac = Dfac_all + (size_t)b * dfac_stride;
  float* y = yall + (size_t)b * batch_stride + (size_t)dr * np;      // residuals: rows of the consumed A buffer, one per draw
  float* nub = nu + (size_t)dr * nu_dstride + (size_t)b * n;
  u += (size_t)dr * u_dstride;
  const int col = c * PW + (tid & (PW - 1)), half = tid >> 7;
  float res = 0.f;
  if (first) {
    if (half == 0 && col < n) res = u[(size_t)col * u_stride + (size_t)b * u_bstride] - factor_elem(Lm, Dfac, np, n + dr, col);
  } else {
    const int r0 = (P + 1) * PW;
    if (tid < PW) sx[tid] = r0 + tid < n ? nub[r0 + tid] : 0.f;
    __syncthreads();
    float acc = 0.f;
#pragma unroll 8
    for (int r = half * (PW / 2); r < (half + 1) * (PW / 2); ++r) acc = fmaf(Lm[(size_t)(r0 + r) * np + col], sx[r], acc);
    if (half == 1) spart[tid - PW] = acc;
    __syncthreads();
    if (half == 0) res = y[col] - (acc + spart[tid]);
  }
  if (c != P) {                                     // workgroup-uniform
    if (half == 0) y[col] = res;
    return;
  }
  // thread t < 128 holds column t of the panel's diagonal block (rows >= t) and residual t
  float colv[PW];
  const int g0 = P * PW;
  if (half == 0) {
#pragma unroll
    for (int r = 0; r < PW; ++r) colv[r] = (r >= tid && g0 + r < n) ? factor_elem(Lm, Dfac, np, g0 + r, g0 + tid) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = PW - 1; r >= 0; --r) {
    if (tid == r) {
      const float x = (g0 + r < n) ? res / colv[r] : 0.f;   // rows past n are padding (row n is the rhs row): x = 0
      sx[r] = x;
      res = x;
    }
    __syncthreads();
    if (tid < r) res = fmaf(-colv[r], sx[r], res);
  }
  if (half == 0 && col < n) nub[col] = res;
}

// nu -> optional dense output and the coefficient fields of the pack's inducing records (as the tail of k_solve_back)
__global__ void k_nu_publish(int n, const float* __restrict__ nu, float* __restrict__ nu_out, int kernel, int Di, int Do,
                             const float* __restrict__ var, float* __restrict__ pack_ind, size_t nu_dstride, size_t pack_dstride) {
  const int b = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  nu += (size_t)blockIdx.z * nu_dstride;
  if (nu_out) nu_out += (size_t)blockIdx.z * nu_dstride;
  pack_ind += (size_t)blockIdx.z * pack_dstride;
  const float v = nu[(size_t)b * n + j];
  if (nu_out) nu_out[(size_t)b * n + j] = v;
  if (!pack_ind) return;
  const int RQ2 = cdiv(Di + Do, 4);
  int m, d;
  float coef;
  if (kernel == 0) { m = j; d = b; coef = var[d] * v; } else { m = j / Do; d = j % Do; coef = v; }
  const int field = Di + d;
  pack_ind[(((size_t)(m >> 6) * RQ2 + (field >> 2)) * 64 + (m & 63)) * 4 + (field & 3)] = coef;
}

// A <- a caller's kernel matrix + jitter I, augmented like k_Kzz: rhs row n = u_prior, huge diagonal there, identity padding.
//   RBF: Ku (Do, M, M), u_prior (M, Do);  DF: Ku (M D, M D), u_prior (M, D) flattened
__global__ void k_fill_from_K(int kernel, int Do, int n, int np, const float* __restrict__ Ku, const float* __restrict__ u_prior,
                              float* __restrict__ A) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y, b = blockIdx.z;
  if (c >= np) return;
  float v;
  if (r < n && c < n) v = Ku[((size_t)b * n + r) * n + c] + (r == c ? kJitter : 0.f);
  else if (r == n) v = c < n ? (kernel == 0 ? u_prior[(size_t)c * Do + b] : u_prior[c]) : (c == n ? 1e30f : 0.f);
  else v = (r == c) ? 1.f : 0.f;
  A[((size_t)b * np + r) * np + c] = v;
}

// dense lower-triangular copy of the factor (zeros above the diagonal)
__global__ void k_copy_L(const float* __restrict__ Aall, const float* __restrict__ Dfac_all, size_t dinv_stride,
                         int n, int np, size_t batch_stride, float* __restrict__ Lu) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y, b = blockIdx.z;
  if (c >= n) return;
  float v = 0.f;
  if (c <= r) {
    const int kb = r / NB;
    v = (c < kb * NB) ? Aall[(size_t)b * batch_stride + (size_t)r * np + c]
                      : Dfac_all[(size_t)b * dinv_stride + ((size_t)kb * NB + (r - kb * NB)) * NB + (c - kb * NB)];
  }
  Lu[((size_t)b * n + r) * n + c] = v;
}

static size_t pack_floats_for(int kernel, int Di, int Do, int M, int S) {
  const size_t SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  if (kernel == 0) return 256 * (SJ * Do * cdiv(Di + 2, 4) + MJ * cdiv(Di + Do, 4)) + (size_t)cdiv(Do * Di, 4) * 4;
  return 256 * (SJ * Do * cdiv(2 * Do + 3, 4) + MJ * cdiv(2 * Do, 4)) + (size_t)cdiv(2 * Do * Do + Do, 4) * 4;
}

int cache_sizes(int kernel, int Di, int Do, int M, int S, size_t* pack_floats, size_t* ws_floats, int nd) {
  if (!dims_supported(kernel, Di, Do)) return set_error("gpode_cache_sizes: no specialisation for kernel=%d Di=%d Do=%d", kernel, Di, Do);
  if (M < 1 || S < 1 || nd < 1) return set_error("gpode_cache_sizes: M=%d S=%d draws=%d", M, S, nd);
  if (pack_floats) *pack_floats = pack_floats_for(kernel, Di, Do, M, S);
  if (ws_floats) *ws_floats = ws_layout(kernel, Di, Do, M, S, nd).total;
  return 0;
}

// Blocked Cholesky of `batch` np x np systems (A is consumed; L goes to Lmat below the diagonal tiles and to Dfac on them).
static void cholesky_blocked(float* A, float* Lmat, float* Dfac, int np, int nblk, int batch, int* info, hipStream_t st, int nreal) {
  struct { int np, nblk, batch; } w{np, nblk, batch};
  const size_t bstride = (size_t)np * np, dstride = (size_t)nblk * NB * NB;
  constexpr int PB = ST / NB;                          // 32-blocks per 128-wide panel
  if (big_factor(w.np)) {
    // big factor (BASELINE configs[4]: 8192 x 8192): panels of 128 columns factored tile column by tile column, then ONE
    // matrix-core rank-128 update of everything to their right -- n^3/3 of the flops on MFMA, and a launch's redundant
    // panel factorisations confined to 4 tile columns instead of the whole trailing triangle
    for (int K = 0; K < w.nblk / PB; ++K) {
      const int jlim = (K + 1) * PB;
      for (int k = K * PB; k < jlim; ++k) {
        int tiles = 0;
        for (int j = k; j < jlim; ++j) tiles += w.nblk - j;
        hipLaunchKernelGGL(k_chol_rl, dim3(tiles, w.batch), 256, 0, st, A, Lmat, w.np, bstride, Dfac, dstride, k, info, w.nblk, jlim, nreal);
      }
      const int Tt = w.nblk / PB - (K + 1);
      if (Tt > 0)
        hipLaunchKernelGGL(k_syrk_mfma, dim3(Tt * (Tt + 1) / 2, w.batch), 256, 0, st, A, Lmat, w.np, bstride, jlim * NB, K * ST);
    }
  } else {
    static const bool single = [] { const char* e = getenv("GPODE_CHOL_SINGLE_STEP"); return e && e[0] == '1'; }();
    int k = 0;
    if (!single)
      for (; k + 1 < w.nblk; k += 2) {                 // two block columns per launch (k_chol_rl2)
        const int T = w.nblk - k;
        hipLaunchKernelGGL(k_chol_rl2, dim3(T * (T + 1) / 2, w.batch), 256, 0, st, A, Lmat, w.np, bstride, Dfac, dstride, k, info, w.nblk, nreal);
      }
    for (; k < w.nblk; ++k) {
      const int T = w.nblk - k;
      hipLaunchKernelGGL(k_chol_rl, dim3(T * (T + 1) / 2, w.batch), 256, 0, st, A, Lmat, w.np, bstride, Dfac, dstride, k, info, w.nblk, w.nblk, nreal);
    }
  }
}

// nd Monte-Carlo draws in one build (nd = 1: SVGP_Layer.build_cache as the reference calls it).  Noise, pack and the per-draw
// outputs (omega, phase, u, nu, u_prior) are stacked along a leading draw axis; ell, var, Lu and the factor in ws are shared.
int cache_build_fwd(int kernel, int Di, int Do, int M, int S, int nd,
                    const float* raw_ell, const float* raw_var, const float* Z, const float* Um, const float* Us_packed,
                    const float* eps_u, const float* rff_w, const float* rff_eps, const float* rff_u,
                    float* pack, float* ws, float* ell, float* var, float* omega, float* phase, float* u,
                    float* Lu, float* nu, float* u_prior, hipStream_t st) {
  if (!dims_supported(kernel, Di, Do)) return set_error("gpode_cache_build_fwd: no specialisation for kernel=%d Di=%d Do=%d", kernel, Di, Do);
  if (kernel == 1 && Di != Do) return set_error("gpode_cache_build_fwd: DF needs D_in == D_out (kernels.py:259-262)");
  if (nd < 1 || nd > 65535) return set_error("gpode_cache_build_fwd: %d draws", nd);
  const WsLayout w = ws_layout(kernel, Di, Do, M, S, nd);
  const size_t SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  const size_t rff_f4 = (kernel == 0 ? SJ * Do * cdiv(Di + 2, 4) : SJ * Do * cdiv(2 * Do + 3, 4)) * 64;
  const size_t ind_f4 = (kernel == 0 ? MJ * cdiv(Di + Do, 4) : MJ * cdiv(2 * Do, 4)) * 64;
  const size_t pf = pack_floats_for(kernel, Di, Do, M, S);
  const size_t MD = (size_t)M * Do;                    // per-draw stride of u, u_prior (and of nu: batch * n = M Do for both kernels)
  float* pack_ind = pack + 4 * rff_f4;
  float* uni = pack + 4 * (rff_f4 + ind_f4);
  int* info = reinterpret_cast<int*>(ws + w.info);
  {
    PrepArgs a;
    a.kernel = kernel; a.Di = Di; a.Do = Do; a.M = M; a.S = S;
    a.raw_ell = raw_ell; a.raw_var = raw_var; a.Z = Z; a.Um = Um; a.Us = Us_packed;
    a.eps_u = eps_u; a.rff_w = rff_w; a.rff_eps = rff_eps; a.rff_u = rff_u;
    a.pack = pack; a.pack_ind = pack_ind; a.uni = uni;
    a.ell_ws = ws + w.ell; a.var_ws = ws + w.var; a.u_ws = ws + w.u;
    a.ell_out = ell; a.var_out = var; a.omega_out = omega; a.phase_out = phase; a.u_out = u;
    a.info = info;
    a.nb_rff = cdiv((int)(SJ * Do * 64), 256);
    a.nb_u = cdiv(M * Do, 4);
    a.nb_ind = cdiv((int)(MJ * 64), 256);
    a.nb_hyp = 1;
    a.nb_om = (omega || phase) ? cdiv(Di * S * Do, 256) : 0;
    a.s_eps_u = MD; a.s_rff_w = (size_t)(kernel == 0 ? S : 2 * S) * Do; a.s_rff_eps = (size_t)Di * S * Do; a.s_rff_u = (size_t)S * Do;
    a.s_pack = pf;
    hipLaunchKernelGGL(k_prep, dim3(a.nb_rff + a.nb_u + a.nb_ind + a.nb_hyp + a.nb_om, nd), 256, 0, st, a);
    if (check_launch("cache prep")) return 1;
  }

  // u_prior = f_prior(Z): the rhs kernel in prior-only mode on the M inducing locations, every draw with its own pack (straight
  // into the caller's buffer when one is given: no copy node on the critical path of the step)
  float* up = u_prior ? u_prior : ws + w.u_prior;
  if (rhs_fwd(kernel, Di, Do, M, S, pack, Z, M, up, 1, st, Draws{nd, pf, 0, MD})) return 1;

  float* A = ws + w.A;
  float* Lmat = ws + w.Lmat;
  float* Dfac = ws + w.Dfac;
  const size_t bstride = (size_t)w.np * w.np, dstride = (size_t)w.nblk * NB * NB;
  if (draw_in_lds(w.np, M, Di, nd)) {
    // small systems: kernel matrix, factorisation and both solves in one launch, LDS-resident
    const size_t lds = draw_lds_bytes(w.np, M, Di, nd);
    if (kernel == 0) {
      if (set_max_lds((const void*)k_draw_lds<0>, lds)) return 1;
      hipLaunchKernelGGL(k_draw_lds<0>, w.batch, 512, lds, st, Di, Do, M, w.n, w.np, w.nblk, Z, ws + w.ell, ws + w.var, up, ws + w.u, Lmat,
                         bstride, Dfac, dstride, ws + w.nu, nu, pack_ind, info, nd, MD, MD, pf);
    } else {
      if (set_max_lds((const void*)k_draw_lds<1>, lds)) return 1;
      hipLaunchKernelGGL(k_draw_lds<1>, w.batch, 512, lds, st, Di, Do, M, w.n, w.np, w.nblk, Z, ws + w.ell, ws + w.var, up, ws + w.u, Lmat,
                         bstride, Dfac, dstride, ws + w.nu, nu, pack_ind, info, nd, MD, MD, pf);
    }
    if (Lu) hipLaunchKernelGGL(k_copy_L, dim3(cdiv(w.n, 128), w.n, w.batch), 128, 0, st, Lmat, Dfac, dstride, w.n, w.np, bstride, Lu);
    return check_launch("cache build (LDS-resident draw)");
  }
  if (kernel == 0)
    hipLaunchKernelGGL(k_Kzz_rbf, dim3(cdiv(w.np, 128), w.np, Do), 128, 0, st, Di, Do, M, w.np, Z, ws + w.ell, ws + w.var, up, A, nd);
  else
    hipLaunchKernelGGL(k_Kzz_df, dim3(cdiv(w.np, 128), w.np, 1), 128, 0, st, Do, M, w.np, Z, ws + w.ell, ws + w.var, up, A, nd);
  cholesky_blocked(A, Lmat, Dfac, w.np, w.nblk, w.batch, info, st, w.n);
  if (check_launch("cholesky")) return 1;

  // nu_l = L^-T (u_l - L^-1 u_prior_l), written to ws, to the optional output and into draw l's pack
  const int u_stride = kernel == 0 ? Do : 1, u_bstride = kernel == 0 ? 1 : 0;
  if (big_factor(w.np)) {
    const int npanel = cdiv(w.n, ST);
    for (int P = npanel - 1; P >= 0; --P)
      hipLaunchKernelGGL(k_solve_back_panel, dim3(P + 1, w.batch, nd), 256, 0, st, Lmat, w.n, w.np, bstride, Dfac, dstride, ws + w.u,
                         u_stride, u_bstride, A, ws + w.nu, P, P == npanel - 1 ? 1 : 0, MD, MD);
    hipLaunchKernelGGL(k_nu_publish, dim3(cdiv(w.n, 256), w.batch, nd), 256, 0, st, w.n, ws + w.nu, nu, kernel, Di, Do, ws + w.var, pack_ind,
                       MD, pf);
  } else {
    if (solve_back_deep_fits(w.np, w.nblk)) {
      const size_t ldsd = sizeof(float) * ((size_t)w.np + (size_t)w.nblk * NB * NB);
      if (set_max_lds((const void*)k_solve_back_deep, ldsd)) return 1;
      hipLaunchKernelGGL(k_solve_back_deep, dim3(w.batch, nd), SBD_THREADS, ldsd, st, Lmat, w.n, w.np, bstride, Dfac, dstride, ws + w.u,
                         u_stride, u_bstride, ws + w.nu, nu, kernel, Di, Do, M, ws + w.var, pack_ind, MD, MD, pf);
      if (Lu) hipLaunchKernelGGL(k_copy_L, dim3(cdiv(w.n, 128), w.n, w.batch), 128, 0, st, Lmat, Dfac, dstride, w.n, w.np, bstride, Lu);
      return check_launch("cache build");
    }
    const size_t lds = sizeof(float) * w.np;
    const int cpt = cdiv(w.n, 1024);  // 4 columns per thread per unit
#define GP_SOLVE(CPT)                                                                                              \
  do {                                                                                                             \
    if (set_max_lds((const void*)k_solve_back<CPT>, lds)) return 1;                                                \
    hipLaunchKernelGGL(k_solve_back<CPT>, dim3(w.batch, nd), 256, lds, st, Lmat, w.n, w.np, bstride, Dfac, dstride, ws + w.u,   \
                       u_stride, u_bstride, ws + w.nu, nu, kernel, Di, Do, M, ws + w.var, pack_ind, MD, MD, pf);   \
  } while (0)
    if (cpt == 1) GP_SOLVE(1);
    else if (cpt == 2) GP_SOLVE(2);
    else GP_SOLVE(0);
#undef GP_SOLVE
  }
  if (Lu) hipLaunchKernelGGL(k_copy_L, dim3(cdiv(w.n, 128), w.n, w.batch), 128, 0, st, Lmat, Dfac, dstride, w.n, w.np, bstride, Lu);
  return check_launch("cache build");
}

// ---------------------------------------------------------------------------------------------
// The kernel's own methods, for a caller that keeps its SVGP_Layer and binds per method (kernels.py:112-137 / :305-316,
// :155-172 / :376-387, :174-181 / :390-393).  They are the stages of cache_build_fwd above, split where the reference splits them:
//   kern_cache   kern.build_cache(S): omega = eps / ell, phase = 2 pi u, and a PRIOR-ONLY pack (M = 0 inducing records) that
//                rhs_fwd(mode 1) evaluates -- kern.rff_forward(x) on the kernel's own Fourier features
//   compute_nu   kern.compute_nu(Ku, u_prior, u): Cholesky of the CALLER's Ku + jitter I and the two triangular solves
//   f_update     kern.f_update(x, x2): K(x, x2) nu with the caller's nu -- an UPDATE-ONLY pack (S = 0) from (x2, nu), rhs_fwd(mode 2)
// scratch: kern_scratch_floats(kernel, Di, Do, M, S) floats = that pack + softplus'd hyper-parameters.
// ---------------------------------------------------------------------------------------------
size_t kern_scratch_floats(int kernel, int Di, int Do, int M, int S) {
  return pack_floats_for(kernel, Di, Do, M, S) + (size_t)cdiv(Do * Di + Do, 4) * 4;
}

static void kern_prep(int kernel, int Di, int Do, int M, int S, const float* raw_ell, const float* raw_var, const float* Z,
                      const float* rff_w, const float* rff_eps, const float* rff_u, float* pack, float* omega, float* phase, hipStream_t st) {
  const size_t SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  const size_t rff_f4 = (kernel == 0 ? SJ * Do * cdiv(Di + 2, 4) : SJ * Do * cdiv(2 * Do + 3, 4)) * 64;
  const size_t ind_f4 = (kernel == 0 ? MJ * cdiv(Di + Do, 4) : MJ * cdiv(2 * Do, 4)) * 64;
  float* hyp = pack + pack_floats_for(kernel, Di, Do, M, S);        // [ell (Do Di) | var (Do)] behind the pack
  PrepArgs a;
  a.kernel = kernel; a.Di = Di; a.Do = Do; a.M = M; a.S = S;
  a.raw_ell = raw_ell; a.raw_var = raw_var; a.Z = Z; a.Um = nullptr; a.Us = nullptr;
  a.eps_u = nullptr; a.rff_w = rff_w; a.rff_eps = rff_eps; a.rff_u = rff_u;
  a.pack = pack; a.pack_ind = pack + 4 * rff_f4; a.uni = pack + 4 * (rff_f4 + ind_f4);
  a.ell_ws = hyp; a.var_ws = hyp + (size_t)Do * Di; a.u_ws = nullptr;
  a.ell_out = nullptr; a.var_out = nullptr; a.omega_out = omega; a.phase_out = phase; a.u_out = nullptr;
  a.info = nullptr;
  a.nb_rff = cdiv((int)(SJ * Do * 64), 256);
  a.nb_u = 0;                                        // no inducing sample here (sample_inducing is the layer's, svpy.py:88-101)
  a.nb_ind = cdiv((int)(MJ * 64), 256);
  a.nb_hyp = 1;
  a.nb_om = (omega || phase) ? cdiv(Di * S * Do, 256) : 0;
  a.s_eps_u = a.s_rff_w = a.s_rff_eps = a.s_rff_u = a.s_pack = 0;
  hipLaunchKernelGGL(k_prep, dim3(a.nb_rff + a.nb_u + a.nb_ind + a.nb_hyp + a.nb_om, 1), 256, 0, st, a);
}

int kern_cache(int kernel, int Di, int Do, int S, const float* raw_ell, const float* raw_var, const float* rff_w, const float* rff_eps,
               const float* rff_u, float* pack, float* omega, float* phase, hipStream_t st) {
  if (!dims_supported(kernel, Di, Do)) return set_error("gpode_kern_cache: no specialisation for kernel=%d Di=%d Do=%d", kernel, Di, Do);
  if (S < 1) return set_error("gpode_kern_cache: S=%d", S);
  kern_prep(kernel, Di, Do, 0, S, raw_ell, raw_var, nullptr, rff_w, rff_eps, rff_u, pack, omega, phase, st);
  return check_launch("kern.build_cache");
}

int compute_nu_ws(int kernel, int Di, int Do, int M, size_t* ws_floats) {
  if (!dims_supported(kernel, Di, Do) || M < 1) return set_error("gpode_compute_nu: kernel=%d Di=%d Do=%d M=%d", kernel, Di, Do, M);
  *ws_floats = ws_layout(kernel, Di, Do, M, 1, 1).total;
  return 0;
}

int compute_nu(int kernel, int Di, int Do, int M, const float* Ku, const float* u_prior, const float* u, float* nu, float* ws, hipStream_t st) {
  size_t need = 0;
  if (compute_nu_ws(kernel, Di, Do, M, &need)) return 1;
  const WsLayout w = ws_layout(kernel, Di, Do, M, 1, 1);
  float* A = ws + w.A;
  float* Lmat = ws + w.Lmat;
  float* Dfac = ws + w.Dfac;
  int* info = reinterpret_cast<int*>(ws + w.info);
  const size_t bstride = (size_t)w.np * w.np, dstride = (size_t)w.nblk * NB * NB, MD = (size_t)M * Do;
  if (hipMemsetAsync(info, 0, 4 * sizeof(int), st) != hipSuccess) return set_error("gpode_compute_nu: memset failed");
  hipLaunchKernelGGL(k_fill_from_K, dim3(cdiv(w.np, 128), w.np, w.batch), 128, 0, st, kernel, Do, w.n, w.np, Ku, u_prior, A);
  cholesky_blocked(A, Lmat, Dfac, w.np, w.nblk, w.batch, info, st, w.n);
  if (check_launch("compute_nu: cholesky")) return 1;
  const int u_stride = kernel == 0 ? Do : 1, u_bstride = kernel == 0 ? 1 : 0;
  if (big_factor(w.np)) {
    const int npanel = cdiv(w.n, ST);
    for (int P = npanel - 1; P >= 0; --P)
      hipLaunchKernelGGL(k_solve_back_panel, dim3(P + 1, w.batch, 1), 256, 0, st, Lmat, w.n, w.np, bstride, Dfac, dstride, u,
                         u_stride, u_bstride, A, ws + w.nu, P, P == npanel - 1 ? 1 : 0, MD, MD);
    hipLaunchKernelGGL(k_nu_publish, dim3(cdiv(w.n, 256), w.batch, 1), 256, 0, st, w.n, ws + w.nu, nu, kernel, Di, Do, (const float*)nullptr,
                       (float*)nullptr, MD, (size_t)0);
  } else {
    if (solve_back_deep_fits(w.np, w.nblk)) {
      const size_t ldsd = sizeof(float) * ((size_t)w.np + (size_t)w.nblk * NB * NB);
      if (set_max_lds((const void*)k_solve_back_deep, ldsd)) return 1;
      hipLaunchKernelGGL(k_solve_back_deep, dim3(w.batch, 1), SBD_THREADS, ldsd, st, Lmat, w.n, w.np, bstride, Dfac, dstride, u, u_stride,
                         u_bstride, ws + w.nu, nu, kernel, Di, Do, M, (const float*)nullptr, (float*)nullptr, MD, MD, (size_t)0);
      return check_launch("kern.compute_nu");
    }
    const size_t lds = sizeof(float) * w.np;
    if (set_max_lds((const void*)k_solve_back<0>, lds)) return 1;
    hipLaunchKernelGGL(k_solve_back<0>, dim3(w.batch, 1), 256, lds, st, Lmat, w.n, w.np, bstride, Dfac, dstride, u, u_stride, u_bstride,
                       ws + w.nu, nu, kernel, Di, Do, M, (const float*)nullptr, (float*)nullptr, MD, MD, (size_t)0);
  }
  return check_launch("kern.compute_nu");
}

int f_update(int kernel, int Di, int Do, int M, const float* raw_ell, const float* raw_var, const float* x2, const float* nu,
             const float* x, int N, float* out, float* pack, hipStream_t st) {
  if (!dims_supported(kernel, Di, Do) || M < 1) return set_error("gpode_f_update: kernel=%d Di=%d Do=%d M=%d", kernel, Di, Do, M);
  kern_prep(kernel, Di, Do, M, 0, raw_ell, raw_var, x2, nullptr, nullptr, nullptr, pack, nullptr, nullptr, st);
  const size_t MJ = cdiv(M, 64);
  (void)MJ;
  float* pack_ind = pack;                             // S = 0: the inducing records open the pack
  const float* var = pack + pack_floats_for(kernel, Di, Do, M, 0) + (size_t)Do * Di;
  const int n = kernel == 0 ? M : M * Do, batch = kernel == 0 ? Do : 1;
  hipLaunchKernelGGL(k_nu_publish, dim3(cdiv(n, 256), batch, 1), 256, 0, st, n, nu, (float*)nullptr, kernel, Di, Do, var, pack_ind,
                     (size_t)0, (size_t)0);
  if (check_launch("kern.f_update: pack")) return 1;
  return rhs_fwd(kernel, Di, Do, M, 0, pack, x, N, out, 2, st);
}

// ---------------------------------------------------------------------------------------------
// SVGP_Layer.build_conditional (svpy.py:176-210), RBF kernel: q(f(x)) = N(m(x), Sigma(x)) in whitened form,
//   A = L^-1 K(Z,x),  m = A^T Um,  Sigma = K(x,x) + A^T (Us Us^T - I) A.
// The triangular solve rides on the factorisation like the rhs row of the cache build: the N rows K(x_n, Z) are appended to
// K_uu (huge diagonal, so that eliminating the appended columns touches nothing), and after the Cholesky rows M.. of the factor
// hold A^T.  Then one workgroup per (query, output) forms t = Us^T a, the mean and the marginal variance
// sigma^2 + |t|^2 - |a|^2; the full covariance pairs the same vectors.
// ---------------------------------------------------------------------------------------------
struct CondLayout { size_t info, A, Lmat, Dfac, a, t, total; int np, nblk; };
static CondLayout cond_layout(int Do, int M, int N) {
  CondLayout c;
  c.nblk = cdiv(M + N, NB);
  if (c.nblk >= 32) c.nblk = (c.nblk + 3) / 4 * 4;
  c.np = c.nblk * NB;
  size_t o = 0;
  auto take = [&](size_t nfl) { size_t at = o; o += (nfl + 3) / 4 * 4; return at; };
  c.info = take(4);
  c.A = take((size_t)Do * c.np * c.np);
  c.Lmat = take((size_t)Do * c.np * c.np);
  c.Dfac = take((size_t)Do * c.nblk * NB * NB);
  c.a = take((size_t)Do * N * M);
  c.t = take((size_t)Do * N * M);
  c.total = o;
  return c;
}

__device__ __forceinline__ float rbf_k(int Di, const float* __restrict__ raw_ell_d, float var_d, const float* __restrict__ p,
                                       const float* __restrict__ q) {
  float s = 0.f;
  for (int i = 0; i < Di; ++i) {
    const float t = (p[i] - q[i]) / softplus_lower(raw_ell_d[i]);
    s = fmaf(t, t, s);
  }
  return var_d * expf(-0.5f * s);
}

__global__ void k_cond_fill(int Di, int Do, int M, int N, int np, const float* __restrict__ raw_ell, const float* __restrict__ raw_var,
                            const float* __restrict__ Z, const float* __restrict__ x, float* __restrict__ A, int* __restrict__ info) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y, d = blockIdx.z;
  if (c == 0 && r == 0 && d == 0) info[0] = 0;
  if (c >= np) return;
  const float vd = softplus_lower(raw_var[d]);
  float v;
  if (r < M) v = c < M ? rbf_k(Di, raw_ell + d * Di, vd, Z + (size_t)r * Di, Z + (size_t)c * Di) + (r == c ? kJitter : 0.f) : 0.f;
  else if (r < M + N) v = c < M ? rbf_k(Di, raw_ell + d * Di, vd, x + (size_t)(r - M) * Di, Z + (size_t)c * Di) : (c == r ? 1e30f : 0.f);
  else v = (r == c) ? 1.f : 0.f;
  A[((size_t)d * np + r) * np + c] = v;
}

// grid (N, Do), block 256, dynamic LDS M floats
__global__ __launch_bounds__(256) void k_cond_rows(int Do, int M, int N, int np, const float* __restrict__ Lall, size_t bstride,
                                                    const float* __restrict__ Dfac_all, size_t dstride,
                                                    const float* __restrict__ raw_var, const float* __restrict__ Um,
                                                    const float* __restrict__ Us, int us_rank1, float* __restrict__ Aws,
                                                    float* __restrict__ Tws, float* __restrict__ mean, float* __restrict__ var_diag) {
  extern __shared__ float sa[];
  __shared__ float red[4][3];
  const int n = blockIdx.x, d = blockIdx.y, tid = threadIdx.x;
  const float* Lm = Lall + (size_t)d * bstride;
  const float* Dfac = Dfac_all + (size_t)d * dstride;
  float* arow = Aws + ((size_t)d * N + n) * M;
  float* trow = Tws + ((size_t)d * N + n) * M;
  for (int m = tid; m < M; m += 256) {
    const float v = factor_elem(Lm, Dfac, np, M + n, m);
    sa[m] = v;
    arow[m] = v;
  }
  __syncthreads();
  // us_rank1 (q_diag=True): the reference multiplies the (M,1) column s = softplus(raw)[:, d] with its transpose (svpy.py:194-195),
  // i.e. Us Us^T = s s^T; t is then the single number s . a, kept in t[0] with zeros behind it
  float acc[3] = {0.f, 0.f, 0.f};                                   // a . Um[:, d],  |t|^2,  |a|^2
  if (us_rank1) {
    const float* sd = Us + (size_t)d * M;
    float part[1] = {0.f}, tot[1];
    for (int j = tid; j < M; j += 256) {
      part[0] = fmaf(sd[j], sa[j], part[0]);
      acc[0] = fmaf(sa[j], Um[(size_t)j * Do + d], acc[0]);
      acc[2] = fmaf(sa[j], sa[j], acc[2]);
      trow[j] = 0.f;
    }
    wave_sum_multi<1>(part, tot);
    if ((tid & 63) == 0) red[tid >> 6][0] = tot[0];
    __syncthreads();
    const float t0 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    __syncthreads();
    if (tid == 0) { trow[0] = t0; acc[1] = t0 * t0; }
  } else {
    const float* Ud = Us + (size_t)d * ((size_t)M * (M + 1) / 2);   // packed lower triangle, row-major (transforms.py:67-69)
    for (int j = tid; j < M; j += 256) {
      float t = 0.f;
      for (int m = j; m < M; ++m) t = fmaf(Ud[(size_t)m * (m + 1) / 2 + j], sa[m], t);   // (Us^T a)_j
      trow[j] = t;
      acc[0] = fmaf(sa[j], Um[(size_t)j * Do + d], acc[0]);
      acc[1] = fmaf(t, t, acc[1]);
      acc[2] = fmaf(sa[j], sa[j], acc[2]);
    }
  }
  float out[3];
  wave_sum_multi<3>(acc, out);
  if ((tid & 63) == 0) { red[tid >> 6][0] = out[0]; red[tid >> 6][1] = out[1]; red[tid >> 6][2] = out[2]; }
  __syncthreads();
  if (tid == 0) {
    const float m0 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    const float t2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    const float a2 = (red[0][2] + red[1][2]) + (red[2][2] + red[3][2]);
    mean[(size_t)n * Do + d] = m0;
    if (var_diag) var_diag[(size_t)n * Do + d] = softplus_lower(raw_var[d]) + (t2 - a2);
  }
}

// cov[n2][n1][d] = k_d(x_n1, x_n2) + t_n1 . t_n2 - a_n1 . a_n2   ((N,N,Do): the transpose the reference returns)
__global__ __launch_bounds__(256) void k_cond_cov(int Di, int Do, int M, int N, const float* __restrict__ raw_ell,
                                                   const float* __restrict__ raw_var, const float* __restrict__ x,
                                                   const float* __restrict__ Aws, const float* __restrict__ Tws, float* __restrict__ cov) {
  const int lane = threadIdx.x & 63;
  const size_t pair = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wavefront per (n1, n2), all outputs d
  if (pair >= (size_t)N * N) return;
  const int n1 = (int)(pair / N), n2 = (int)(pair % N);
  for (int d = 0; d < Do; ++d) {
    const float* a1 = Aws + ((size_t)d * N + n1) * M;
    const float* a2 = Aws + ((size_t)d * N + n2) * M;
    const float* t1 = Tws + ((size_t)d * N + n1) * M;
    const float* t2 = Tws + ((size_t)d * N + n2) * M;
    float acc = 0.f;
    for (int m = lane; m < M; m += 64) acc += t1[m] * t2[m] - a1[m] * a2[m];
    const float in1[1] = {acc};
    float out1[1];
    wave_sum_multi<1>(in1, out1);
    if (lane == 0)
      cov[((size_t)n2 * N + n1) * Do + d] = rbf_k(Di, raw_ell + d * Di, softplus_lower(raw_var[d]), x + (size_t)n1 * Di, x + (size_t)n2 * Di) + out1[0];
  }
}

int conditional_ws(int Di, int Do, int M, int N, size_t* ws_floats) {
  if (Di <= 0 || Do <= 0 || M <= 0 || N <= 0) return set_error("gpode_conditional: Di=%d Do=%d M=%d N=%d", Di, Do, M, N);
  *ws_floats = cond_layout(Do, M, N).total;
  return 0;
}

int conditional(int Di, int Do, int M, int N, const float* raw_ell, const float* raw_var, const float* Z, const float* Um,
                const float* Us_packed, int us_rank1, const float* x, int full_cov, float* mean, float* var, float* ws, hipStream_t st) {
  size_t need = 0;
  if (conditional_ws(Di, Do, M, N, &need)) return 1;
  const CondLayout c = cond_layout(Do, M, N);
  int* info = reinterpret_cast<int*>(ws + c.info);
  const size_t bstride = (size_t)c.np * c.np, dstride = (size_t)c.nblk * NB * NB;
  hipLaunchKernelGGL(k_cond_fill, dim3(cdiv(c.np, 128), c.np, Do), 128, 0, st, Di, Do, M, N, c.np, raw_ell, raw_var, Z, x, ws + c.A, info);
  cholesky_blocked(ws + c.A, ws + c.Lmat, ws + c.Dfac, c.np, c.nblk, Do, info, st, M);
  if (check_launch("conditional: cholesky")) return 1;
  const size_t lds = sizeof(float) * M;
  if (set_max_lds((const void*)k_cond_rows, lds)) return 1;
  hipLaunchKernelGGL(k_cond_rows, dim3(N, Do), 256, lds, st, Do, M, N, c.np, ws + c.Lmat, bstride, ws + c.Dfac, dstride, raw_var, Um,
                     Us_packed, us_rank1, ws + c.a, ws + c.t, mean, full_cov ? nullptr : var);
  if (full_cov)
    hipLaunchKernelGGL(k_cond_cov, (unsigned)(((size_t)N * N + 3) / 4), 256, 0, st, Di, Do, M, N, raw_ell, raw_var, x, ws + c.a, ws + c.t, var);
  return check_launch("conditional");
}

}  // namespace gp

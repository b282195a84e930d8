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
#include "gp_eval.hpp"
#include "gp_launch.hpp"

namespace gp {

static constexpr int NB = 32;
static constexpr float kJitter = 1e-5f;  // kernels.py:11

__device__ __forceinline__ float softplus_lower(float x) {
  // F.softplus(x) + 1e-12  (constraint_utils.py:5-7; torch threshold 20)
  float sp = x > 20.f ? x : log1pf(expf(x));
  return sp + 1e-12f;
}

// ---------------------------------------------------------------------------------------------
// hyper-parameters + uniform tail
// ---------------------------------------------------------------------------------------------
__global__ void k_hyper(int kernel, int Di, int Do, const float* __restrict__ raw_ell, const float* __restrict__ raw_var,
                        float* __restrict__ ell_ws, float* __restrict__ var_ws, float* __restrict__ ell_out,
                        float* __restrict__ var_out, float* __restrict__ uni) {
  const int t = threadIdx.x;
  for (int e = t; e < Do * Di; e += blockDim.x) {
    float l = softplus_lower(raw_ell[e]);
    ell_ws[e] = l;
    if (ell_out) ell_out[e] = l;
    float il2 = 1.f / (l * l);
    if (kernel == 0) uni[e] = -0.5f * GP_LOG2E * il2;        // wl[d][i]
    else { uni[e] = -0.5f * GP_LOG2E * il2; uni[Do * Di + e] = il2; }  // wab[a][b], il2[a][b]
  }
  for (int d = t; d < Do; d += blockDim.x) {
    float v = softplus_lower(raw_var[d]);
    var_ws[d] = v;
    if (var_out) var_out[d] = v;
    if (kernel == 1) uni[2 * Do * Di + d] = v;
  }
}

// omega[i,s,d] = eps[i,s,d] / ell[d,i];  phase[s,d] = u * 2 * pi  (API-visible copies)
__global__ void k_omega(int Di, int Do, int S, const float* __restrict__ rff_eps, const float* __restrict__ rff_u,
                        const float* __restrict__ ell, float* __restrict__ omega_ws, float* __restrict__ omega_out,
                        float* __restrict__ phase_out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int tot = Di * S * Do;
  if (e < tot) {
    const int d = e % Do, i = e / (S * Do);
    float o = rff_eps[e] / ell[d * Di + i];
    omega_ws[e] = o;
    if (omega_out) omega_out[e] = o;
  }
  if (phase_out && e < S * Do) phase_out[e] = (rff_u[e] * 2.f) * 3.14159265358979323846f;
}

// u[n,d] = sum_{m<=n} Us[d, n(n+1)/2 + m] eps_u[m,d] + Um[n,d]   (svpy.py:94-100, transforms.py:71-77)
__global__ void k_inducing_sample(int M, int Do, const float* __restrict__ Us, const float* __restrict__ eps_u,
                                  const float* __restrict__ Um, float* __restrict__ u_ws, float* __restrict__ u_out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= M * Do) return;
  const int n = e / Do, d = e % Do;
  const float* row = Us + (size_t)d * ((size_t)M * (M + 1) / 2) + (size_t)n * (n + 1) / 2;
  float acc = 0.f;
  for (int m = 0; m <= n; ++m) acc = fmaf(row[m], eps_u[m * Do + d], acc);
  acc += Um[e];
  u_ws[e] = acc;
  if (u_out) u_out[e] = acc;
}

__device__ __forceinline__ void put_rec(float* __restrict__ pack, size_t rec_f4_base, int RQ, int lane, int field, float v) {
  // float index of (record base in float4 units, quad q = field/4, lane, component field%4)
  pack[((rec_f4_base + (size_t)(field >> 2)) * 64 + lane) * 4 + (field & 3)] = v;
}

// RBF rff records: thread per (j,d,lane)
__global__ void k_pack_rff_rbf(int Di, int Do, int S, const float* __restrict__ omega, const float* __restrict__ rff_u,
                               const float* __restrict__ rff_w, const float* __restrict__ var, float* __restrict__ pack) {
  const int RQ = cdiv(Di + 2, 4);
  const int SJ = cdiv(S, 64);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= SJ * Do * 64) return;
  const int lane = e & 63, d = (e >> 6) % Do, j = (e >> 6) / Do;
  const int s = j * 64 + lane;
  const size_t base = (size_t)(j * Do + d) * RQ;
  const bool ok = s < S;
  for (int i = 0; i < Di; ++i) put_rec(pack, base, RQ, lane, i, ok ? omega[((size_t)i * S + s) * Do + d] * GP_INV2PI : 0.f);
  put_rec(pack, base, RQ, lane, Di, ok ? rff_u[s * Do + d] : 0.f);
  put_rec(pack, base, RQ, lane, Di + 1, ok ? sqrtf(var[d] / (float)S) * rff_w[s * Do + d] : 0.f);
  for (int f = Di + 2; f < 4 * RQ; ++f) put_rec(pack, base, RQ, lane, f, 0.f);
}

// DF rff records: thread per (j,i,lane)
__global__ void k_pack_rff_df(int D, int S, const float* __restrict__ omega, const float* __restrict__ rff_u,
                              const float* __restrict__ rff_w, const float* __restrict__ var, float* __restrict__ pack) {
  const int RQ = cdiv(2 * D + 3, 4);
  const int SJ = cdiv(S, 64);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= SJ * D * 64) return;
  const int lane = e & 63, i = (e >> 6) % D, j = (e >> 6) / D;
  const int s = j * 64 + lane;
  const size_t base = (size_t)(j * D + i) * RQ;
  const bool ok = s < S;
  // theta_si = u[s,i] + sum_k x_k omega[k,s,i]
  for (int k = 0; k < D; ++k) put_rec(pack, base, RQ, lane, k, ok ? omega[((size_t)k * S + s) * D + i] * GP_INV2PI : 0.f);
  put_rec(pack, base, RQ, lane, D, ok ? rff_u[s * D + i] : 0.f);
  put_rec(pack, base, RQ, lane, D + 1, ok ? rff_w[s * D + i] : 0.f);
  put_rec(pack, base, RQ, lane, D + 2, ok ? rff_w[(S + s) * D + i] : 0.f);
  // B[s,i,jj] = norm_{s,jj} delta_{i,jj} - (sum_k omega[i,s,k] omega[jj,s,k]) / norm_{s,jj},
  // norm_{s,jj} = sqrt(sum_k' omega[k',s,jj]^2)   (kernels.py:327-336)
  for (int jj = 0; jj < D; ++jj) {
    float v = 0.f;
    if (ok) {
      float n2 = 0.f, g = 0.f;
      for (int k = 0; k < D; ++k) {
        float o = omega[((size_t)k * S + s) * D + jj];
        n2 = fmaf(o, o, n2);
        g = fmaf(omega[((size_t)i * S + s) * D + k], omega[((size_t)jj * S + s) * D + k], g);
      }
      float nrm = sqrtf(n2);
      v = ((i == jj) ? nrm : 0.f) - g / nrm;
      v *= sqrtf(var[jj] / (float)S);
    }
    put_rec(pack, base, RQ, lane, D + 3 + jj, v);
  }
  for (int f = 2 * D + 3; f < 4 * RQ; ++f) put_rec(pack, base, RQ, lane, f, 0.f);
}

// inducing records: thread per (j,lane).  nu == nullptr -> coefficient fields are zero.
__global__ void k_pack_ind(int kernel, int Di, int Do, int M, const float* __restrict__ Z, const float* __restrict__ nu,
                           const float* __restrict__ var, float* __restrict__ pack_ind) {
  const int RQ2 = cdiv(Di + Do, 4);
  const int MJ = cdiv(M, 64);
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= MJ * 64) return;
  const int lane = e & 63, j = e >> 6;
  const int m = j * 64 + lane;
  const size_t base = (size_t)j * RQ2;
  const bool ok = m < M;
  for (int i = 0; i < Di; ++i) put_rec(pack_ind, base, RQ2, lane, i, ok ? Z[m * Di + i] : 0.f);
  for (int d = 0; d < Do; ++d) {
    float v = 0.f;
    if (ok && nu) v = (kernel == 0) ? var[d] * nu[(size_t)d * M + m] : nu[(size_t)m * Do + d];
    put_rec(pack_ind, base, RQ2, lane, Di + d, v);
  }
  for (int f = Di + Do; f < 4 * RQ2; ++f) put_rec(pack_ind, base, RQ2, lane, f, 0.f);
}

// ---------------------------------------------------------------------------------------------
// K(Z) + jitter I, augmented with the rhs row (row n) and identity padding up to np.
//   A: (batch, np, np) row-major.  RBF: batch = Do, n = M.  DF: batch = 1, n = M D.
// ---------------------------------------------------------------------------------------------
__global__ void k_Kzz_rbf(int Di, int Do, int M, int np, const float* __restrict__ Z, const float* __restrict__ ell,
                          const float* __restrict__ var, const float* __restrict__ u_prior, float* __restrict__ A) {
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
  } else if (r == M) {
    v = c < M ? u_prior[c * Do + d] : (c == M ? 1e30f : 0.f);
  } else {
    v = (r == c) ? 1.f : 0.f;
  }
  A[((size_t)d * np + r) * np + c] = v;
}

__global__ void k_Kzz_df(int D, int M, int np, const float* __restrict__ Z, const float* __restrict__ ell,
                         const float* __restrict__ var, const float* __restrict__ u_prior, float* __restrict__ A) {
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
  } else if (r == n) {
    v = c < n ? u_prior[c] : (c == n ? 1e30f : 0.f);
  } else {
    v = (r == c) ? 1.f : 0.f;
  }
  A[(size_t)r * np + c] = v;
}

// ---------------------------------------------------------------------------------------------
// 32x32 Cholesky / triangular inverse in the registers of one wavefront.
// Lane r (lanes 32..63 mirror lanes 0..31) holds row r.  Only entries c <= r are meaningful.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void chol32_wave(float (&row)[NB], int r, int* __restrict__ info) {
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    float piv = __shfl(row[j], j, 64);
    if (!(piv > 0.f) && r == 0 && info) atomicOr(info, 1);  // not positive definite (reference raises)
    float s = sqrtf(piv);
    float inv = 1.f / s;
    row[j] = (r == j) ? s : row[j] * inv;
#pragma unroll
    for (int c = j + 1; c < NB; ++c) {
      float lcj = __shfl(row[j], c, 64);
      row[c] = fmaf(-row[j], lcj, row[c]);
    }
  }
}

// lane c gets column c of L^-1: x[r] = Linv[r][c]
__device__ __forceinline__ void trinv32_wave(const float (&Lrow)[NB], float (&x)[NB], int c) {
#pragma unroll
  for (int r = 0; r < NB; ++r) {
    float acc = (r == c) ? 1.f : 0.f;
#pragma unroll
    for (int p = 0; p < r; ++p) {
      float l = __shfl(Lrow[p], r, 64);
      acc = fmaf(-l, x[p], acc);
    }
    float d = __shfl(Lrow[r], r, 64);
    x[r] = acc / d;
  }
}

// One block column k of the left-looking factorisation.  grid = (nblk - k, batch), block = 256.
// The factored diagonal block L_kk goes to Dfac (NOT in place: the other workgroups of this launch still
// read the unfactored A_kk), its inverse to Dinv; off-diagonal panel blocks are overwritten in place.
__global__ __launch_bounds__(256) void k_chol_step(float* __restrict__ Aall, int np, size_t batch_stride,
                                                    float* __restrict__ Dinv_all, float* __restrict__ Dfac_all,
                                                    size_t dinv_stride, int k, int* __restrict__ info) {
  __shared__ float sLk[NB][NB + 1];
  __shared__ float sLi[NB][NB + 1];
  __shared__ float sD[NB][NB + 1];    // diagonal block -> L_kk^-1
  __shared__ float sR[NB][NB + 1];    // this workgroup's panel block
  float* A = Aall + (size_t)blockIdx.y * batch_stride;
  float* Dinv = Dinv_all + (size_t)blockIdx.y * dinv_stride + (size_t)k * NB * NB;
  float* Dfac = Dfac_all + (size_t)blockIdx.y * dinv_stride + (size_t)k * NB * NB;
  const int i = k + blockIdx.x;
  const int row0 = i * NB, col0 = k * NB;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  float accD[4] = {0.f, 0.f, 0.f, 0.f}, accR[4] = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < k; ++p) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q;
      sLk[r][tx] = A[(size_t)(col0 + r) * np + p * NB + tx];
      sLi[r][tx] = A[(size_t)(row0 + r) * np + p * NB + tx];
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < NB; ++c) {
      const float b = sLk[tx][c];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        accD[q] = fmaf(sLk[ty + 8 * q][c], b, accD[q]);
        accR[q] = fmaf(sLi[ty + 8 * q][c], b, accR[q]);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q;
    sD[r][tx] = A[(size_t)(col0 + r) * np + col0 + tx] - accD[q];
    sR[r][tx] = A[(size_t)(row0 + r) * np + col0 + tx] - accR[q];
  }
  __syncthreads();
  if (tid < 64) {
    const int r = tid & 31;
    float row[NB], x[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) row[c] = sD[r][c];
    chol32_wave(row, r, (blockIdx.x == 0) ? info : nullptr);
    trinv32_wave(row, x, r);
    if (tid < 32) {
      if (i == k) {  // the diagonal workgroup publishes L_kk (lower) and L_kk^-1
#pragma unroll
        for (int c = 0; c < NB; ++c) Dfac[r * NB + c] = (c <= r) ? row[c] : 0.f;
#pragma unroll
        for (int rr = 0; rr < NB; ++rr) Dinv[rr * NB + r] = x[rr];
      }
#pragma unroll
      for (int rr = 0; rr < NB; ++rr) sD[rr][r] = x[rr];  // sD <- L_kk^-1 (row rr, col r)
    }
  }
  __syncthreads();
  if (i != k) {
    // X = R L_kk^-T :  X[r][c] = sum_{p<=c} R[r][p] Linv[c][p]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q;
      float acc = 0.f;
      for (int p = 0; p <= tx; ++p) acc = fmaf(sR[r][p], sD[tx][p], acc);
      A[(size_t)(row0 + r) * np + col0 + tx] = acc;
    }
  }
}

// nu = L^-T (u - y),  y = row n of the factor (forward-solved rhs).  grid = batch, block = 256.
//   u element (j) of batch b at u[j * u_stride + b * u_bstride]; nu written dense (batch, n).
__global__ __launch_bounds__(256) void k_solve_back(const float* __restrict__ Aall, int n, int np, size_t batch_stride,
                                                     const float* __restrict__ Dinv_all, const float* __restrict__ Dfac_all,
                                                     size_t dinv_stride,
                                                     const float* __restrict__ u, int u_stride, int u_bstride,
                                                     float* __restrict__ nu) {
  extern __shared__ float sv[];  // np floats: residual, overwritten by the solution
  __shared__ float sx[NB];
  const float* A = Aall + (size_t)blockIdx.x * batch_stride;
  const float* Dinv = Dinv_all + (size_t)blockIdx.x * dinv_stride;
  const int tid = threadIdx.x;
  // y = row n of the factor: left of the last diagonal block it sits in A, inside it in Dfac
  const int kl = n / NB, cl = kl * NB;
  const float* Dl = Dfac_all + (size_t)blockIdx.x * dinv_stride + (size_t)kl * NB * NB;
  for (int j = tid; j < np; j += blockDim.x) {
    float y = 0.f;
    if (j < n) y = (j < cl) ? A[(size_t)n * np + j] : Dl[(n - cl) * NB + (j - cl)];
    sv[j] = j < n ? u[(size_t)j * u_stride + (size_t)blockIdx.x * u_bstride] - y : 0.f;
  }
  __syncthreads();
  const int nblk = cdiv(n, NB);
  for (int k = nblk - 1; k >= 0; --k) {
    const int c0 = k * NB;
    if (tid < NB) {
      // x_k[c] = sum_{r>=c, c0+r<n} Linv_kk[r][c] * res[c0+r]
      const float* Li = Dinv + (size_t)k * NB * NB;
      float acc = 0.f;
      for (int r = tid; r < NB; ++r)
        if (c0 + r < n) acc = fmaf(Li[r * NB + tid], sv[c0 + r], acc);
      sx[tid] = (c0 + tid < n) ? acc : 0.f;
    }
    __syncthreads();
    if (tid < NB) sv[c0 + tid] = sx[tid];
    // res[c] -= sum_r L[c0+r][c] x_k[r]  for c < c0
    for (int c = tid; c < c0; c += blockDim.x) {
      float acc = sv[c];
#pragma unroll 8
      for (int r = 0; r < NB; ++r)
        if (c0 + r < n) acc = fmaf(-A[(size_t)(c0 + r) * np + c], sx[r], acc);
      sv[c] = acc;
    }
    __syncthreads();
  }
  for (int j = tid; j < n; j += blockDim.x) nu[(size_t)blockIdx.x * n + j] = sv[j];
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

// ---------------------------------------------------------------------------------------------
// workspace layout (floats)
// ---------------------------------------------------------------------------------------------
struct WsLayout {
  size_t info, ell, var, omega, u, u_prior, nu, A, Dinv, Dfac, total;
  int n, np, nblk, batch;
};

static WsLayout ws_layout(int kernel, int Di, int Do, int M, int S) {
  WsLayout w;
  w.n = kernel == 0 ? M : M * Do;
  w.batch = kernel == 0 ? Do : 1;
  w.nblk = cdiv(w.n + 1, NB);
  w.np = w.nblk * NB;
  size_t o = 0;
  auto take = [&](size_t nfl) { size_t at = o; o += (nfl + 3) / 4 * 4; return at; };
  w.info = take(4);
  w.ell = take((size_t)Do * Di);
  w.var = take(Do);
  w.omega = take((size_t)Di * S * Do);
  w.u = take((size_t)M * Do);
  w.u_prior = take((size_t)M * Do);
  w.nu = take((size_t)w.batch * w.n);
  w.A = take((size_t)w.batch * w.np * w.np);
  w.Dinv = take((size_t)w.batch * w.nblk * NB * NB);
  w.Dfac = take((size_t)w.batch * w.nblk * NB * NB);
  w.total = o;
  return w;
}

static size_t pack_floats_for(int kernel, int Di, int Do, int M, int S) {
  const size_t SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  if (kernel == 0) return 256 * (SJ * Do * cdiv(Di + 2, 4) + MJ * cdiv(Di + Do, 4)) + (size_t)cdiv(Do * Di, 4) * 4;
  return 256 * (SJ * Do * cdiv(2 * Do + 3, 4) + MJ * cdiv(2 * Do, 4)) + (size_t)cdiv(2 * Do * Do + Do, 4) * 4;
}

int cache_sizes(int kernel, int Di, int Do, int M, int S, size_t* pack_floats, size_t* ws_floats) {
  if (!dims_supported(kernel, Di, Do)) return set_error("gpode_cache_sizes: no specialisation for kernel=%d Di=%d Do=%d", kernel, Di, Do);
  if (M < 1 || S < 1) return set_error("gpode_cache_sizes: M=%d S=%d", M, S);
  if (pack_floats) *pack_floats = pack_floats_for(kernel, Di, Do, M, S);
  if (ws_floats) *ws_floats = ws_layout(kernel, Di, Do, M, S).total;
  return 0;
}

int cache_build_fwd(int kernel, int Di, int Do, int M, int S,
                    const float* raw_ell, const float* raw_var, const float* Z, const float* Um, const float* Us_packed,
                    const float* eps_u, const float* rff_w, const float* rff_eps, const float* rff_u,
                    float* pack, float* ws, float* ell, float* var, float* omega, float* phase, float* u,
                    float* Lu, float* nu, float* u_prior, hipStream_t st) {
  if (!dims_supported(kernel, Di, Do)) return set_error("gpode_cache_build_fwd: no specialisation for kernel=%d Di=%d Do=%d", kernel, Di, Do);
  if (kernel == 1 && Di != Do) return set_error("gpode_cache_build_fwd: DF needs D_in == D_out (kernels.py:259-262)");
  const WsLayout w = ws_layout(kernel, Di, Do, M, S);
  const size_t SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  const size_t rff_f4 = (kernel == 0 ? SJ * Do * cdiv(Di + 2, 4) : SJ * Do * cdiv(2 * Do + 3, 4)) * 64;
  const size_t ind_f4 = (kernel == 0 ? MJ * cdiv(Di + Do, 4) : MJ * cdiv(2 * Do, 4)) * 64;
  float* pack_ind = pack + 4 * rff_f4;
  float* uni = pack + 4 * (rff_f4 + ind_f4);
  int* info = reinterpret_cast<int*>(ws + w.info);
  hipError_t e = hipMemsetAsync(info, 0, 16, st);
  if (e != hipSuccess) return set_error("memset: %s", hipGetErrorString(e));

  hipLaunchKernelGGL(k_hyper, 1, 256, 0, st, kernel, Di, Do, raw_ell, raw_var, ws + w.ell, ws + w.var, ell, var, uni);
  {
    const int tot = Di * S * Do;
    hipLaunchKernelGGL(k_omega, cdiv(tot, 256), 256, 0, st, Di, Do, S, rff_eps, rff_u, ws + w.ell, ws + w.omega, omega, phase);
  }
  hipLaunchKernelGGL(k_inducing_sample, cdiv(M * Do, 128), 128, 0, st, M, Do, Us_packed, eps_u, Um, ws + w.u, u);
  if (kernel == 0)
    hipLaunchKernelGGL(k_pack_rff_rbf, cdiv((int)(SJ * Do * 64), 256), 256, 0, st, Di, Do, S, ws + w.omega, rff_u, rff_w, ws + w.var, pack);
  else
    hipLaunchKernelGGL(k_pack_rff_df, cdiv((int)(SJ * Do * 64), 256), 256, 0, st, Do, S, ws + w.omega, rff_u, rff_w, ws + w.var, pack);
  hipLaunchKernelGGL(k_pack_ind, cdiv((int)(MJ * 64), 256), 256, 0, st, kernel, Di, Do, M, Z, (const float*)nullptr, ws + w.var, pack_ind);
  if (check_launch("cache prep")) return 1;

  // u_prior = f_prior(Z): the rhs kernel in prior-only mode on the M inducing locations
  if (rhs_fwd(kernel, Di, Do, M, S, pack, Z, M, ws + w.u_prior, 1, st)) return 1;
  if (u_prior) {
    e = hipMemcpyAsync(u_prior, ws + w.u_prior, sizeof(float) * M * Do, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return set_error("memcpy: %s", hipGetErrorString(e));
  }

  float* A = ws + w.A;
  float* Dinv = ws + w.Dinv;
  float* Dfac = ws + w.Dfac;
  const size_t bstride = (size_t)w.np * w.np, dstride = (size_t)w.nblk * NB * NB;
  if (kernel == 0)
    hipLaunchKernelGGL(k_Kzz_rbf, dim3(cdiv(w.np, 128), w.np, Do), 128, 0, st, Di, Do, M, w.np, Z, ws + w.ell, ws + w.var, ws + w.u_prior, A);
  else
    hipLaunchKernelGGL(k_Kzz_df, dim3(cdiv(w.np, 128), w.np, 1), 128, 0, st, Do, M, w.np, Z, ws + w.ell, ws + w.var, ws + w.u_prior, A);
  for (int k = 0; k < w.nblk; ++k)
    hipLaunchKernelGGL(k_chol_step, dim3(w.nblk - k, w.batch), 256, 0, st, A, w.np, bstride, Dinv, Dfac, dstride, k, info);
  if (check_launch("cholesky")) return 1;

  // nu = L^-T (u - L^-1 u_prior)
  {
    const int u_stride = kernel == 0 ? Do : 1, u_bstride = kernel == 0 ? 1 : 0;
    const size_t lds = sizeof(float) * w.np;
    if (set_max_lds((const void*)k_solve_back, lds)) return 1;
    hipLaunchKernelGGL(k_solve_back, w.batch, 256, lds, st, A, w.n, w.np, bstride, Dinv, Dfac, dstride, ws + w.u, u_stride, u_bstride, ws + w.nu);
  }
  hipLaunchKernelGGL(k_pack_ind, cdiv((int)(MJ * 64), 256), 256, 0, st, kernel, Di, Do, M, Z, ws + w.nu, ws + w.var, pack_ind);
  if (nu) {
    e = hipMemcpyAsync(nu, ws + w.nu, sizeof(float) * w.batch * w.n, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return set_error("memcpy: %s", hipGetErrorString(e));
  }
  if (Lu) hipLaunchKernelGGL(k_copy_L, dim3(cdiv(w.n, 128), w.n, w.batch), 128, 0, st, A, Dfac, dstride, w.n, w.np, bstride, Lu);
  return check_launch("cache build");
}

}  // namespace gp

// gp_cache_bwd.hip -- backward of the per-draw cache build: pack-layout gradients -> raw parameters.
//
// What autograd does in the reference behind SVGP_Layer.build_cache (svpy.py:103-121) when
// loss.backward() runs (main.py:210): gradients flow from every use of omega / variance / nu / Z / ell
// inside the 4(T-1) RHS evaluations back through  nu = L^-T (u - L^-1 f_prior(Z)),  L = chol(K(Z)+jitter I),
// u = tril(Us) eps + Um,  omega = eps/ell,  softplus.  Here:
//
//   g_nu  <- coefficient fields of the pack gradient (gp_backward.hip, kernel B)
//   q = L^-1 g_nu,  a = L^-T q (= K^-1 g_nu),  v = L^-1 p,  r = u - v          (matvecs with L^-1)
//   g_u = q   -> g_Um, g_Us ;   g_p = -a  -> f_prior(Z) path: VJP w.r.t. Z + parameter gradients (prior only)
//   g_L = -tril(nu q^T) + tril(a v^T)  =>  Phi = tril_half(L^T g_L) = tril_half(-r q^T + q v^T)
//   g_K = sym(L^-T Phi L^-1)            (torch's cholesky_backward convention: symmetrised, both triangles)
//   g_K -> g_Z, g_ell, g_var through the kernel-matrix formula (kernels.py:98-110 / :289-303)
//   pack chain: om = eps/(2 pi ell), aw = sqrt(var/S) w, cc = var nu, wl = -log2e/(2 ell^2), B(omega) ...
//   raw chain: softplus' = sigmoid.
//
// L^-1 is formed explicitly (all block columns in parallel, one launch) so that everything after it is
// plain tiled FMAs with no sequential chain; cond(L) = sqrt(cond(K)) ~ 1e2, harmless in fp32.
#include "gp_eval.hpp"
#include "gp_launch.hpp"
#include "gp_ws.hpp"
#include "wave_reduce.hpp"

namespace gp {

__device__ __forceinline__ float sigmoid_raw(float x) { return x > 20.f ? 1.f : 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float softplus_l(float x) { return (x > 20.f ? x : log1pf(expf(x))) + 1e-12f; }

struct BwsLayout {
  size_t Dinv, Linv, X, S, vec, gp_rows, gu_rows, vjpZ, slab, gZpart, kpart, gom, total;
  int n, np, nbn, batch, nchunkZ, kpart_stride, nd;
};

// nd Monte-Carlo draws share the factor (ws_layout): the vectors, the f_prior(Z) path and the pack chains are per draw, the
// triangular inverse / solves, Phi (summed over the draws: the Cholesky backward is linear in it) and the K(Z) backward are not.
static BwsLayout bws_layout(int kernel, int Di, int Do, int M, int S, size_t pack_floats, int nd) {
  BwsLayout b;
  const WsLayout w = ws_layout(kernel, Di, Do, M, S, nd);
  b.nd = nd;
  b.n = w.n; b.np = w.np; b.batch = w.batch; b.nbn = cdiv(w.n, NB);
  b.nchunkZ = M < 16 ? M : 16;
  size_t o = 0;
  auto take = [&](size_t nfl) { size_t at = o; o += (nfl + 3) / 4 * 4; return at; };
  b.Dinv = take((size_t)b.batch * b.nbn * NB * NB);
  b.Linv = take((size_t)b.batch * b.np * b.np);
  b.X = take((size_t)b.batch * b.np * b.np);
  b.S = take((size_t)b.batch * b.np * b.np);
  b.vec = take((size_t)6 * nd * b.batch * b.np);
  b.gp_rows = take((size_t)nd * M * Do);
  b.gu_rows = take((size_t)nd * M * Do);
  b.vjpZ = take((size_t)nd * M * Di);
  b.slab = take((size_t)nd * b.nchunkZ * pack_floats);
  b.gZpart = take((size_t)(kernel == 0 ? Do : 1) * M * Di);
  b.kpart_stride = kernel == 0 ? (1 + Di) : (Do + Do * Do);
  b.kpart = take((size_t)(kernel == 0 ? Do * M : M) * b.kpart_stride);
  b.gom = take(kernel == 1 ? (size_t)nd * S * Do * Do : 4);
  b.total = o;
  return b;
}

// vector k (0 g_nu, 1 q, 2 a, 3 r, 4 v) of draw l, system b:  [k][draw][system][np]
#define GP_VEC(base, k, l) ((base) + ((size_t)((k) * nd + (l)) * nb + b) * np)

// ---------------------------------------------------------------------------------------------
// g_nu from the coefficient fields of the pack gradient
// ---------------------------------------------------------------------------------------------
__global__ void k_gnu(int kernel, int Di, int Do, int M, int n, int np, const float* __restrict__ gpack_ind,
                      const float* __restrict__ var, float* __restrict__ vec_all, size_t pack_dstride) {
  const int b = blockIdx.y, nb = gridDim.y, l = blockIdx.z, nd = gridDim.z;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= np) return;
  gpack_ind += (size_t)l * pack_dstride;
  float* gnu = GP_VEC(vec_all, 0, l);
  float v = 0.f;
  if (j < n) {
    const int RQ2 = cdiv(Di + Do, 4);
    int m, d;
    if (kernel == 0) { m = j; d = b; } else { m = j / Do; d = j % Do; }
    const int field = Di + d;
    const float gc = gpack_ind[(((size_t)(m >> 6) * RQ2 + (field >> 2)) * 64 + (m & 63)) * 4 + (field & 3)];
    v = kernel == 0 ? gc * var[d] : gc;
  }
  gnu[j] = v;
}

// ---------------------------------------------------------------------------------------------
// inverse of every 32x32 diagonal block of L (rows/cols >= n treated as identity)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_trinv_diag(const float* __restrict__ Dfac_all, size_t dfac_stride, int n,
                                                    float* __restrict__ Dinv_all, size_t dinv_stride) {
  __shared__ float sL[NB][NB + 1];
  const int k = blockIdx.x, b = blockIdx.y, c0 = k * NB;
  const float* Lk = Dfac_all + (size_t)b * dfac_stride + (size_t)k * NB * NB;
  float* out = Dinv_all + (size_t)b * dinv_stride + (size_t)k * NB * NB;
  for (int e = threadIdx.x; e < NB * NB; e += 64) {
    const int r = e / NB, c = e % NB;
    sL[r][c] = (c0 + r < n && c0 + c < n) ? Lk[e] : (r == c ? 1.f : 0.f);
  }
  __syncthreads();
  const int c = threadIdx.x;
  if (c < NB) {
    float x[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
      float acc = (r == c) ? 1.f : 0.f;
#pragma unroll
      for (int p = 0; p < r; ++p) acc = fmaf(-sL[r][p], x[p], acc);
      x[r] = acc / sL[r][r];
    }
#pragma unroll
    for (int r = 0; r < NB; ++r) out[r * NB + c] = x[r];
  }
}

// ---------------------------------------------------------------------------------------------
// L^-1 by divide and conquer on the block structure:  [A 0; B C]^-1 = [A^-1 0; -C^-1 B A^-1  C^-1].
// Level l merges neighbouring super-blocks of s = 2^l blocks (32 s rows); every level is two batched tile GEMMs
// (T = B A^-1, then X21 = -C^-1 T) over all pairs, so the whole inverse is 2 ceil(log2 nbn) launches of wide
// grids instead of a forward substitution whose critical path is nbn^2 / 2 dependent tile products.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_linv_init(int np, const float* __restrict__ Dinv_all, size_t dinv_stride,
                                                    float* __restrict__ Linv_all, size_t batch_stride) {
  const int i = blockIdx.y, j = blockIdx.x, b = blockIdx.z;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  float* Li = Linv_all + (size_t)b * batch_stride;
  const float* Dinv = Dinv_all + (size_t)b * dinv_stride + (size_t)i * NB * NB;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = ty + 8 * q;
    Li[(size_t)(i * NB + r) * np + j * NB + tx] = (i == j) ? Dinv[r * NB + tx] : 0.f;
  }
}

// step 0: T[C,A] = L[C,A] Linv[A,A]      (Linv[A,A] lower triangular: k-blocks >= column block)
// step 1: Linv[C,A] = -Linv[C,C] T[C,A]  (Linv[C,C] lower triangular: k-blocks <= row block)
// grid (sb, sb, npairs * batch); A = blocks [2 sb pair, +sb), C = the following <= sb blocks
__global__ __launch_bounds__(256) void k_linv_dc(int step, int sb, int npairs, int nbn, int n, int np, const float* __restrict__ Lall,
                                                  float* __restrict__ Linv_all, float* __restrict__ T_all, size_t batch_stride) {
  __shared__ float sA[NB][NB + 1], sB[NB][NB + 1];
  const int pair = blockIdx.z % npairs, b = blockIdx.z / npairs;
  const int a0 = 2 * sb * pair, c0 = a0 + sb;
  if (c0 >= nbn) return;
  const int cs = min(sb, nbn - c0), rb = blockIdx.y, cb = blockIdx.x;
  if (rb >= cs) return;
  const float* Lm = Lall + (size_t)b * batch_stride;
  float* Li = Linv_all + (size_t)b * batch_stride;
  float* T = T_all + (size_t)b * batch_stride;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int row0 = (c0 + rb) * NB, col0 = (a0 + cb) * NB;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const int k_lo = step == 0 ? cb : 0, k_hi = step == 0 ? sb : rb + 1;
  const float* P = step == 0 ? Lm : Li;
  const float* Q = step == 0 ? Li : T;
  float ra[4], rq[4];                                // next k-block's tiles, in flight during the current product
  auto load = [&](int kb) {
    const int kcol = (step == 0 ? a0 + kb : c0 + kb) * NB;           // P columns / Q rows
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q;
      ra[q] = (step == 1 || row0 + r < n) ? P[(size_t)(row0 + r) * np + kcol + tx] : 0.f;   // rows >= n of L are not part of the factor
      rq[q] = Q[(size_t)(kcol + r) * np + col0 + tx];
    }
  };
  if (k_lo < k_hi) load(k_lo);
  for (int kb = k_lo; kb < k_hi; ++kb) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) { sA[ty + 8 * q][tx] = ra[q]; sB[ty + 8 * q][tx] = rq[q]; }
    __syncthreads();
    if (kb + 1 < k_hi) load(kb + 1);
#pragma unroll 8
    for (int p = 0; p < NB; ++p) {
      const float bv = sB[p][tx];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = fmaf(sA[ty + 8 * q][p], bv, acc[q]);
    }
  }
  float* out = step == 0 ? T : Li;
#pragma unroll
  for (int q = 0; q < 4; ++q) out[(size_t)(row0 + ty + 8 * q) * np + col0 + tx] = step == 0 ? acc[q] : -acc[q];
}

// ---------------------------------------------------------------------------------------------
// vectors: q = L^-1 g_nu, a = L^-T q, v = L^-1 p (row n of the factor), r = u - v ; g_Um ; g_p rows
//   vec layout per batch: [gnu | q | a | r | v | -] each np floats.  One wavefront per output element.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_vec_q(int n, int np, const float* __restrict__ Linv_all, size_t batch_stride,
                                               float* __restrict__ vec_all) {
  const int b = blockIdx.y, nb = gridDim.y, l = blockIdx.z, nd = gridDim.z, lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= np) return;
  const float* Li = Linv_all + (size_t)b * batch_stride + (size_t)i * np;
  const float* vg = GP_VEC(vec_all, 0, l);
  float acc = 0.f;
  if (i < n)
    for (int j = lane; j <= i; j += 64) acc = fmaf(Li[j], vg[j], acc);
  const float in1[1] = {acc};
  float out1[1];
  wave_sum_multi<1>(in1, out1);
  if (lane == 0) GP_VEC(vec_all, 1, l)[i] = out1[0];
}

__global__ __launch_bounds__(256) void k_vec_a(int kernel, int Do, int n, int np, const float* __restrict__ Lall, size_t batch_stride,
                                               const float* __restrict__ Dfac_all, size_t dfac_stride, const float* __restrict__ Linv_all,
                                               const float* __restrict__ u, float* __restrict__ vec_all, float* __restrict__ gu_rows,
                                               float* __restrict__ gp_rows, size_t u_dstride) {
  const int b = blockIdx.y, nb = gridDim.y, l = blockIdx.z, nd = gridDim.z, lane = threadIdx.x & 63;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= np) return;
  const float* Lm = Lall + (size_t)b * batch_stride;
  const float* Li = Linv_all + (size_t)b * batch_stride;
  const float* Dfac = Dfac_all + (size_t)b * dfac_stride;
  const float* vq = GP_VEC(vec_all, 1, l);
  u += (size_t)l * u_dstride; gu_rows += (size_t)l * u_dstride; gp_rows += (size_t)l * u_dstride;
  float acc = 0.f;
  if (j < n)
    for (int i = j + lane; i < n; i += 64) acc = fmaf(Li[(size_t)i * np + j], vq[i], acc);
  const float in1[1] = {acc};
  float out1[1];
  wave_sum_multi<1>(in1, out1);
  if (lane != 0) return;
  const float a = out1[0];
  float y = 0.f, rr = 0.f;
  if (j < n) {
    const int rn = n + l, kl = rn / NB, cl = kl * NB;           // draw l's forward-solved rhs is row n + l of the factor
    const int u_stride = kernel == 0 ? Do : 1, u_b = kernel == 0 ? b : 0;
    y = (j < cl) ? Lm[(size_t)rn * np + j] : Dfac[(size_t)kl * NB * NB + (rn - cl) * NB + (j - cl)];
    rr = u[(size_t)j * u_stride + u_b] - y;
    // g_u = q (u element j of batch b lives at u[j*u_stride + u_b]); g_p = -a in the same (M,Do) layout
    gu_rows[(size_t)j * u_stride + u_b] = vq[j];
    gp_rows[(size_t)j * u_stride + u_b] = -a;
  }
  GP_VEC(vec_all, 2, l)[j] = a;
  GP_VEC(vec_all, 3, l)[j] = rr;
  GP_VEC(vec_all, 4, l)[j] = y;
}

// g_Us[d, n(n+1)/2 + m] = sum_l g_u_l[n,d] eps_u_l[m,d]   (svpy.py:94-100 backward);  g_Um = sum_l g_u_l.  Draws in a fixed order.
// accum: g_Um / g_Us already hold a gradient (the KL term's, written earlier on another path) and the flow's is added to it
__global__ void k_gUs(int M, int Do, int nd, const float* __restrict__ g_u, const float* __restrict__ eps_u, float* __restrict__ g_Us,
                      float* __restrict__ g_Um, int accum) {
  const size_t P = (size_t)M * (M + 1) / 2, MD = (size_t)M * Do;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < MD) {
    float acc = 0.f;
    for (int l = 0; l < nd; ++l) acc += g_u[l * MD + e];
    g_Um[e] = accum ? g_Um[e] + acc : acc;
  }
  if (e >= P * Do) return;
  const int d = (int)(e / P);
  const size_t k = e % P;
  int nn = (int)((sqrtf(8.f * (float)k + 1.f) - 1.f) * 0.5f);
  while ((size_t)(nn + 1) * (nn + 2) / 2 <= k) ++nn;
  while ((size_t)nn * (nn + 1) / 2 > k) --nn;
  const int m = (int)(k - (size_t)nn * (nn + 1) / 2);
  float acc = 0.f;
  for (int l = 0; l < nd; ++l) acc = fmaf(g_u[l * MD + (size_t)nn * Do + d], eps_u[l * MD + (size_t)m * Do + d], acc);
  g_Us[e] = accum ? g_Us[e] + acc : acc;
}

// The solve route's form of k_gUs: g_u = q and g_p = -a are rows 1 and 2 of the vector table (element (m, d) of draw l: system b = d,
// entry m for the RBF kernel; the single system, entry m Do + d for the divergence-free one), so the launch that spread them into
// (M, Do) rows (k_vec_rv's second half) is folded in: this kernel writes gp_rows itself and reads g_u where it lies.
__global__ void k_gUs_vec(int kernel, int M, int Do, int nd, int nb, int np, const float* __restrict__ vec_all,
                          const float* __restrict__ eps_u, float* __restrict__ gp_rows, float* __restrict__ g_Us, float* __restrict__ g_Um,
                          int accum) {
  const size_t P = (size_t)M * (M + 1) / 2, MD = (size_t)M * Do;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  auto at = [&](int k, int l, int m, int d) {
    const int b = kernel == 0 ? d : 0;
    const size_t j = kernel == 0 ? (size_t)m : (size_t)m * Do + d;
    return GP_VEC(vec_all, k, l)[j];
  };
  if (e < MD) {
    const int m = (int)(e / Do), d = (int)(e % Do);
    float acc = 0.f;
    for (int l = 0; l < nd; ++l) {
      acc += at(1, l, m, d);
      gp_rows[l * MD + e] = -at(2, l, m, d);
    }
    g_Um[e] = accum ? g_Um[e] + acc : acc;
  }
  if (e >= P * Do) return;
  const int d = (int)(e / P);
  const size_t k = e % P;
  int nn = (int)((sqrtf(8.f * (float)k + 1.f) - 1.f) * 0.5f);
  while ((size_t)(nn + 1) * (nn + 2) / 2 <= k) ++nn;
  while ((size_t)nn * (nn + 1) / 2 > k) --nn;
  const int m = (int)(k - (size_t)nn * (nn + 1) / 2);
  float acc = 0.f;
  for (int l = 0; l < nd; ++l) acc = fmaf(at(1, l, nn, d), eps_u[l * MD + (size_t)m * Do + d], acc);
  g_Us[e] = accum ? g_Us[e] + acc : acc;
}

// Phi[i][j] = sum_l (-r_l[i] q_l[j] + q_l[i] v_l[j]): the Cholesky backward is linear in Phi, so the draws that share the factor are
// summed HERE, in front of the two triangular products (vectors laid out GP_VEC: [k][draw][system][np])
__device__ __forceinline__ float phi_sum(const float* __restrict__ vec_all, int nd, int nb, int b, int np, size_t i, size_t j) {
  float ph = 0.f;
  for (int l = 0; l < nd; ++l) {
    const float* vq = GP_VEC(vec_all, 1, l);
    ph += -GP_VEC(vec_all, 3, l)[i] * vq[j] + vq[i] * GP_VEC(vec_all, 4, l)[j];
  }
  return ph;
}

// ---------------------------------------------------------------------------------------------
// X = L^-T Phi,  Phi[i][j] = -r_i q_j + q_i v_j (i > j), half of that on the diagonal, 0 above
// 32 x 32 output tile per workgroup, the tile products on the matrix cores: wavefront w owns the 16 x 16 quadrant
// (w & 1, w >> 1), 8 v_mfma_f32_16x16x4_f32 per 32-deep k tile (exact fp32; the VALU version spent ~1.5 us per k tile on 160 LDS
// reads per thread -- 29 + 33 us for the two products of the 608-row factor, on the exposed tail of the step).
// ---------------------------------------------------------------------------------------------
typedef float tf32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_gemm_phiX(const float* __restrict__ Linv_all, size_t batch_stride, int np, int nbn,
                                                    const float* __restrict__ vec_all, float* __restrict__ X_all, int nd) {
  __shared__ float sA[NB][NB + 1], sP[NB][NB + 1];
  const int ab = blockIdx.x, jb = blockIdx.y, b = blockIdx.z, nb = gridDim.z;
  const float* Li = Linv_all + (size_t)b * batch_stride;
  float* X = X_all + (size_t)b * batch_stride;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int lane = tid & 63, lr = lane & 15, lk = lane >> 4, mq = (tid >> 6) & 1, nq = tid >> 7;
  tf32x4 macc = tf32x4{0.f, 0.f, 0.f, 0.f};
  float ra[4], rp[4];                                                // next block: Linv tile and the Phi tile generated on the fly
  auto load = [&](int ib) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = ty + 8 * q, gi = ib * NB + p, gj = jb * NB + tx;
      ra[q] = Li[(size_t)gi * np + ab * NB + tx];                    // Linv[i][a]
      const float ph = gi >= gj ? phi_sum(vec_all, nd, nb, b, np, gi, gj) : 0.f;
      rp[q] = gi > gj ? ph : (gi == gj ? 0.5f * ph : 0.f);
    }
  };
  const int ib0 = ab > jb ? ab : jb;
  load(ib0);
  for (int ib = ib0; ib < nbn; ++ib) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) { sA[ty + 8 * q][tx] = ra[q]; sP[ty + 8 * q][tx] = rp[q]; }
    __syncthreads();
    if (ib + 1 < nbn) load(ib + 1);
    // D[m = a][n = j] += sum_p Linv[p][a] Phi[p][j]:  A[m][k] = sA[k][m],  B[k][n] = sP[k][n]
#pragma unroll
    for (int ks = 0; ks < NB / 4; ++ks)
      macc = __builtin_amdgcn_mfma_f32_16x16x4f32(sA[4 * ks + lk][16 * mq + lr], sP[4 * ks + lk][16 * nq + lr], macc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) X[(size_t)(ab * NB + 16 * mq + 4 * lk + r) * np + jb * NB + 16 * nq + lr] = macc[r];
}

// S = X L^-1
__global__ __launch_bounds__(256) void k_gemm_S(const float* __restrict__ X_all, const float* __restrict__ Linv_all,
                                                 size_t batch_stride, int np, int nbn, float* __restrict__ S_all) {
  __shared__ float sA[NB][NB + 1], sB[NB][NB + 1];
  const int ab = blockIdx.x, cb = blockIdx.y, b = blockIdx.z;
  const float* X = X_all + (size_t)b * batch_stride;
  const float* Li = Linv_all + (size_t)b * batch_stride;
  float* Sm = S_all + (size_t)b * batch_stride;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int lane = tid & 63, lr = lane & 15, lk = lane >> 4, mq = (tid >> 6) & 1, nq = tid >> 7;
  tf32x4 macc = tf32x4{0.f, 0.f, 0.f, 0.f};
  // tiles of k-block jb + 1 travel to registers while block jb is multiplied (the loop is latency-, not bandwidth-bound)
  float ra[4], rb[4];
  auto load = [&](int jb) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int r = ty + 8 * q;
      ra[q] = X[(size_t)(ab * NB + r) * np + jb * NB + tx];
      rb[q] = Li[(size_t)(jb * NB + r) * np + cb * NB + tx];
    }
  };
  load(cb);
  for (int jb = cb; jb < nbn; ++jb) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) { sA[ty + 8 * q][tx] = ra[q]; sB[ty + 8 * q][tx] = rb[q]; }
    __syncthreads();
    if (jb + 1 < nbn) load(jb + 1);
    // D[m = r][n = c] += sum_p X[r][p] Linv[p][c]:  A[m][k] = sA[m][k],  B[k][n] = sB[k][n]
#pragma unroll
    for (int ks = 0; ks < NB / 4; ++ks)
      macc = __builtin_amdgcn_mfma_f32_16x16x4f32(sA[16 * mq + lr][4 * ks + lk], sB[4 * ks + lk][16 * nq + lr], macc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) Sm[(size_t)(ab * NB + 16 * mq + 4 * lk + r) * np + cb * NB + 16 * nq + lr] = macc[r];
}

// ---------------------------------------------------------------------------------------------
// The same two products on the matrix cores for big factors (np a multiple of 128: the 8192 x 8192 K_uu of BASELINE
// configs[4] is 2 x 0.55 TFLOP here).  128 x 128 output tile per workgroup, 4 wavefronts x (4 x 4) v_mfma_f32_16x16x4_f32
// tiles, k in steps of 16 through LDS ([k][128 + 16]: the four k rows of an operand fetch fall on disjoint banks), the next
// k-tile travelling to registers while the current one is multiplied.  MODE 0: S = X Linv (k >= column tile);
// MODE 1: X = Linv^T tril(Phi), Phi generated from its two rank-1 terms while staging (k >= both tiles).
// ---------------------------------------------------------------------------------------------
typedef float gf32x4 __attribute__((ext_vector_type(4)));
constexpr int GT = 128, GK = 16, GLD = GT + 16;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_gemm_mfma(const float* __restrict__ A_all, const float* __restrict__ B_all,
                                                       size_t batch_stride, int np, int klim,
                                                       const float* __restrict__ vec_all, float* __restrict__ C_all, int nd) {
  __shared__ __attribute__((aligned(16))) float sA[GK][GLD], sB[GK][GLD];
  const int tm = blockIdx.x, tn = blockIdx.y, b = blockIdx.z, nb = gridDim.z;
  const float* A = A_all + (size_t)b * batch_stride;
  const float* B = B_all + (size_t)b * batch_stride;
  float* C = C_all + (size_t)b * batch_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  gf32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = gf32x4{0.f, 0.f, 0.f, 0.f};
  const int nkt = klim / GK;                         // L^-1 is defined on the first klim (= 32 nbn) rows only
  const int kt0 = (MODE == 0 ? tn : (tm > tn ? tm : tn)) * (GT / GK);
  float4 ra[2], rb[2];
  auto load = [&](int kt) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 256 * q;
      const int k = e >> 5, c4 = (e & 31) * 4;                      // row k of the k-tile, 4 consecutive columns
      const size_t gk = (size_t)kt * GK + k;
      if (MODE == 0) {
        const int m = e >> 2, k4 = (e & 3) * 4;                     // A = X[a][k]: k is the contiguous index
        ra[q] = *reinterpret_cast<const float4*>(&A[((size_t)tm * GT + m) * np + (size_t)kt * GK + k4]);
        rb[q] = *reinterpret_cast<const float4*>(&B[gk * np + (size_t)tn * GT + c4]);
      } else {
        ra[q] = *reinterpret_cast<const float4*>(&A[gk * np + (size_t)tm * GT + c4]);   // Linv[i][a]: a contiguous
        float ph[4] = {0.f, 0.f, 0.f, 0.f};
        const size_t gj0 = (size_t)tn * GT + c4;
        if (gk >= gj0) {                                              // a quad wholly above the diagonal stays zero
          for (int l = 0; l < nd; ++l) {
            const float* vq = GP_VEC(vec_all, 1, l);
            const float* vv = GP_VEC(vec_all, 4, l);
            const float ri = GP_VEC(vec_all, 3, l)[gk], qi = vq[gk];
#pragma unroll
            for (int c = 0; c < 4; ++c) ph[c] += -ri * vq[gj0 + c] + qi * vv[gj0 + c];
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) ph[c] = gk > gj0 + c ? ph[c] : (gk == gj0 + c ? 0.5f * ph[c] : 0.f);
        }
        rb[q] = make_float4(ph[0], ph[1], ph[2], ph[3]);
      }
    }
  };
  load(kt0);
  for (int kt = kt0; kt < nkt; ++kt) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 256 * q;
      const int k = e >> 5, c4 = (e & 31) * 4;
      if (MODE == 0) {
        const int m = e >> 2, k4 = (e & 3) * 4;
        sA[k4 + 0][m] = ra[q].x; sA[k4 + 1][m] = ra[q].y; sA[k4 + 2][m] = ra[q].z; sA[k4 + 3][m] = ra[q].w;
      } else {
        *reinterpret_cast<float4*>(&sA[k][c4]) = ra[q];
      }
      *reinterpret_cast<float4*>(&sB[k][c4]) = rb[q];
    }
    __syncthreads();
    if (kt + 1 < nkt) load(kt + 1);
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
      float af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { af[i] = sA[4 * ks + lk][wm + 16 * i + lr]; bf[i] = sB[4 * ks + lk][wn + 16 * i + lr]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
  // lane: rows wm + 16 i + 4 lk + r, column wn + 16 j + lr
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[((size_t)tm * GT + wm + 16 * i + 4 * lk + r) * np + (size_t)tn * GT + wn + 16 * j + lr] = acc[i][j][r];
}

// The two divide-and-conquer steps on the matrix cores for super-blocks of >= 128 rows (sb % 4 == 0, 128-aligned): the tile
// engine of k_gemm_mfma with sub-block operands.  grid (sb / 4, sb / 4, npairs * batch).
__global__ __launch_bounds__(256, 2) void k_linv_dc_mfma(int step, int sb, int npairs, int nbn, int n, int np,
                                                          const float* __restrict__ Lall, float* __restrict__ Linv_all,
                                                          float* __restrict__ T_all, size_t batch_stride) {
  __shared__ __attribute__((aligned(16))) float sA[GK][GLD], sB[GK][GLD];
  const int pair = blockIdx.z % npairs, b = blockIdx.z / npairs;
  const int a0 = 2 * sb * pair, c0 = a0 + sb;
  if (c0 >= nbn) return;
  const int cs = min(sb, nbn - c0), tm = blockIdx.y, tn = blockIdx.x;
  if (tm * (GT / NB) >= cs) return;
  const float* P = (step == 0 ? Lall : Linv_all) + (size_t)b * batch_stride;
  const float* Q = (step == 0 ? Linv_all : T_all) + (size_t)b * batch_stride;
  float* out = (step == 0 ? T_all : Linv_all) + (size_t)b * batch_stride;
  const int row0 = c0 * NB + tm * GT, col0 = a0 * NB + tn * GT;
  const int kbase = (step == 0 ? a0 : c0) * NB;                       // P columns / Q rows of k = 0
  const int klo = step == 0 ? tn * GT : 0;                            // Linv[A,A] lower triangular: k >= column tile
  const int khi = step == 0 ? sb * NB : min((tm + 1) * GT, cs * NB);  // Linv[C,C] lower triangular: k <= row tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
  gf32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = gf32x4{0.f, 0.f, 0.f, 0.f};
  float4 ra[2], rb[2];
  auto load = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 256 * q;
      const int m = e >> 2, k4 = (e & 3) * 4, k = e >> 5, c4 = (e & 31) * 4;
      const bool live = step == 1 || row0 + m < n;                    // rows >= n of L are not part of the factor
      ra[q] = live ? *reinterpret_cast<const float4*>(&P[(size_t)(row0 + m) * np + kbase + k0 + k4]) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[q] = *reinterpret_cast<const float4*>(&Q[(size_t)(kbase + k0 + k) * np + col0 + c4]);
    }
  };
  if (klo < khi) load(klo);
  for (int k0 = klo; k0 < khi; k0 += GK) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = tid + 256 * q;
      const int m = e >> 2, k4 = (e & 3) * 4, k = e >> 5, c4 = (e & 31) * 4;
      sA[k4 + 0][m] = ra[q].x; sA[k4 + 1][m] = ra[q].y; sA[k4 + 2][m] = ra[q].z; sA[k4 + 3][m] = ra[q].w;
      *reinterpret_cast<float4*>(&sB[k][c4]) = rb[q];
    }
    __syncthreads();
    if (k0 + GK < khi) load(k0 + GK);
#pragma unroll
    for (int ks = 0; ks < GK / 4; ++ks) {
      float af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { af[i] = sA[4 * ks + lk][wm + 16 * i + lr]; bf[i] = sB[4 * ks + lk][wn + 16 * i + lr]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        out[(size_t)(row0 + wm + 16 * i + 4 * lk + r) * np + col0 + wn + 16 * j + lr] = step == 0 ? acc[i][j][r] : -acc[i][j][r];
}



// ---------------------------------------------------------------------------------------------
// Solve-based route (np <= kTrsmMaxNp): what torch's autograd does behind cholesky / triangular_solve (kernels.py:163-171,
// :384-386) -- triangular SOLVES with the factor, not products with an explicit L^-1.  The explicit inverse (k_linv_dc above)
// is built from products of inverted sub-blocks and loses accuracy with cond(L): on a numerically rank-deficient K_uu (1056
// inducing points in a 3-D latent) d/d ell came out 12 % from fp64 where torch's fp32 solves stay at 1 %
// (tools/ab_bigfactor.py).  Here only the 32 x 32 DIAGONAL blocks are inverted (k_trinv_diag; their condition is that of a
// block, not of the factor) and everything else is block substitution:
//     q = L^-1 g_nu            (forward)        MODE 0, one slab per system (column 0)
//     [X | a] = L^-T [Phi | q] (backward)       MODE 1, slabs of 32 columns of Phi = tril_half(-r q^T + q v^T) + one slab for q
//     S^T = L^-T X^T           (backward)       MODE 2, slabs of 32 rows of X  (S = X L^-1)
// A triangular solve is sequential along the rows but independent per right-hand-side column, so ONE workgroup owns a slab of
// 32 columns for the whole substitution: the slab (np x 32) lives in LDS, there is no inter-workgroup synchronisation, and
// the n^2/2 x 32 multiply-adds per slab run on the matrix cores (v_mfma_f32_16x16x4_f32; the factor's tiles stream from L2
// straight into MFMA operands, each element used once per slab).  Per block step: Y_k = Dinv_k(^T) R_k, then
// R_j -= L_(k,j)^T Y_k for the blocks still to be solved (tiles dealt round-robin to the 4 wavefronts).
// ---------------------------------------------------------------------------------------------
constexpr int TSW = 16;                              // right-hand-side columns per slab (one MFMA tile column)
constexpr int TSL = TSW + 1;                         // slab row stride in LDS (floats)
static constexpr int kTrsmMaxNp = 1216;              // beyond: the explicit-inverse matrix-core route (BASELINE configs[4])

constexpr int TNW = 8;                               // wavefronts per slab workgroup

template <int MODE>
__global__ __launch_bounds__(64 * TNW) void k_trsm_slab(int n, int np, int nbn, const float* __restrict__ Lall, size_t batch_stride,
                                                         const float* __restrict__ Dfac_all, size_t dfac_stride,
                                                         const float* __restrict__ Dinv_all, size_t dinv_stride,
                                                         float* __restrict__ vec_all, const float* __restrict__ Xin_all,
                                                         float* __restrict__ Out_all, int nd) {
  extern __shared__ __attribute__((aligned(16))) float sY[];        // [32 nbn][TSL] slab, then two [32][TSL] scratch tiles
  constexpr bool TRANS = MODE != 0;
  constexpr int NT = 64 * TNW;
  const int slab = blockIdx.x, b = blockIdx.y, nb = gridDim.y;
  const float* Lm = Lall + (size_t)b * batch_stride;
  const float* Dfac = Dfac_all + (size_t)b * dfac_stride;
  const float* Dinv = Dinv_all + (size_t)b * dinv_stride;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lk = lane >> 4;
  // MODE 0: slab s carries g_nu of the draws 16 s .. 16 s + 15 (one column each); MODE 1: slabs >= nslab carry their q likewise
  const int nrow = nbn * NB, c0 = slab * TSW, nslab = (nrow + TSW - 1) / TSW;
  const int l0 = (MODE == 0 ? slab : slab - nslab) * TSW;            // first draw of a vector slab
  float* sT0 = sY + (size_t)nrow * TSL;              // first solution y0 of the diagonal step
  float* sT1 = sT0 + NB * TSL;                       // its residual
  // ---- right-hand sides -------------------------------------------------------------------
  for (int e = tid; e < nrow * TSW; e += NT) {
    float v = 0.f;
    if (MODE == 0) {
      const int i = e / TSW, c = e % TSW;
      if (l0 + c < nd && i < n) v = GP_VEC(vec_all, 0, l0 + c)[i];
      sY[i * TSL + c] = v;
    } else if (MODE == 1) {
      const int i = e / TSW, c = e % TSW, gj = c0 + c;
      if (slab >= nslab) { if (l0 + c < nd && i < n) v = GP_VEC(vec_all, 1, l0 + c)[i]; }
      else if (i < n && gj < n && i >= gj) {
        const float ph = phi_sum(vec_all, nd, nb, b, np, i, gj);
        v = i > gj ? ph : 0.5f * ph;
      }
      sY[i * TSL + c] = v;
    } else {                                         // B[i][c] = X[c0 + c][i]: walk X's rows (i contiguous)
      const int c = e / nrow, i = e - c * nrow, gj = c0 + c;
      if (i < n && gj < n) v = Xin_all[(size_t)b * batch_stride + (size_t)gj * np + i];
      sY[i * TSL + c] = v;
    }
  }
  // ---- block substitution ------------------------------------------------------------------
  // Diagonal step, wavefronts 0 / 1 (rows 0..15 / 16..31 of the block): y0 = Dinv^(T) r, then ONE refinement with the block of the
  // factor itself, y = y0 + Dinv^(T) (r - L_kk^(T) y0).  Dinv alone carries an error of eps * cond(L_kk) -- on a rank-deficient
  // K_uu the pivots of a block range from 1 down to sqrt(jitter) -- the refined solution is as good as a substitution with L_kk.
  auto loadD = [&](int kb, float (&df)[NB / 4], float (&lf)[NB / 4]) {   // A operands for row block mb = wave & 1
    const float* Dk = Dinv + (size_t)kb * NB * NB;
    const float* Lk = Dfac + (size_t)kb * NB * NB;
    const int m = (wave & 1) * 16 + lr;
#pragma unroll
    for (int ks = 0; ks < NB / 4; ++ks) {
      const int k = 4 * ks + lk;
      df[ks] = TRANS ? Dk[k * NB + m] : Dk[m * NB + k];
      const int rr = TRANS ? k : m, cc = TRANS ? m : k;                  // element (rr, cc) of L_kk (lower triangle)
      const bool in = kb * NB + rr < n && kb * NB + cc < n;
      lf[ks] = in ? (cc <= rr ? Lk[rr * NB + cc] : 0.f) : (rr == cc ? 1.f : 0.f);   // rows / columns >= n: identity, as k_trinv_diag
    }
  };
  auto loadA = [&](int kb, int t, float (&af)[2][NB / 4]) {   // tile t of the step: TRANS: L[kb][j]^T, j = t; else L[j][kb], j = kb + 1 + t
    const int j = TRANS ? t : kb + 1 + t;
#pragma unroll
    for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
      for (int ks = 0; ks < NB / 4; ++ks) {
        const int k = 4 * ks + lk, m = m2 * 16 + lr;
        const int row = TRANS ? kb * NB + k : j * NB + m, col = TRANS ? j * NB + m : kb * NB + k;
        af[m2][ks] = (row < n && col < n) ? Lm[(size_t)row * np + col] : 0.f;   // rows / columns >= n: not part of the factor
      }
  };
  // 32 x 16 product of this wavefront's 16 rows: A operand af (8 k-steps), B rows from an LDS tile; two chains over k
  auto mm = [&](const float (&af)[NB / 4], const float* __restrict__ Bt) {
    gf32x4 a0 = gf32x4{0.f, 0.f, 0.f, 0.f}, a1 = gf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NB / 8; ++ks) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks], Bt[(4 * ks + lk) * TSL + lr], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[NB / 8 + ks], Bt[(4 * (NB / 8 + ks) + lk) * TSL + lr], a1, 0, 0, 0);
    }
    return a0 + a1;
  };
  float dfn[NB / 4], lfn[NB / 4], afA[2][NB / 4], afB[2][NB / 4];
  {
    const int kb0 = TRANS ? nbn - 1 : 0;
    if (wave < 2) loadD(kb0, dfn, lfn);
    if (wave < (TRANS ? kb0 : nbn - 1 - kb0)) loadA(kb0, wave, afA);
  }
  __syncthreads();
  for (int step = 0; step < nbn; ++step) {
    const int kb = TRANS ? nbn - 1 - step : step;
    float* Rk = sY + (size_t)kb * NB * TSL;
    const int r0 = (wave & 1) * 16 + 4 * lk;         // this lane's accumulator rows r0 .. r0 + 3, column lr
    gf32x4 y0;
    if (wave < 2) {
      y0 = mm(dfn, Rk);
#pragma unroll
      for (int r = 0; r < 4; ++r) sT0[(r0 + r) * TSL + lr] = y0[r];
    }
    __syncthreads();
    if (wave < 2) {
      const gf32x4 p = mm(lfn, sT0);                 // L_kk^(T) y0
#pragma unroll
      for (int r = 0; r < 4; ++r) sT1[(r0 + r) * TSL + lr] = Rk[(r0 + r) * TSL + lr] - p[r];
    }
    __syncthreads();                                 // residual complete; R_k no longer needed
    if (wave < 2) {
      const gf32x4 d = mm(dfn, sT1);
#pragma unroll
      for (int r = 0; r < 4; ++r) Rk[(r0 + r) * TSL + lr] = y0[r] + d[r];
    }
    // the next step's diagonal-block operands travel under this step's update
    const int kbn = TRANS ? kb - 1 : kb + 1;
    if (wave < 2 && step + 1 < nbn) loadD(kbn, dfn, lfn);
    __syncthreads();
    // R_j -= A_j Y_k for the blocks still to be solved: TRANS: j < kb, A_j = L[kb][j]^T; else j > kb, A_j = L[j][kb]
    const int ntile = TRANS ? kb : nbn - 1 - kb;
    if (ntile > 0) {
      float bf[NB / 4];                              // Y_k as B operand, shared by all tiles of the step
#pragma unroll
      for (int ks = 0; ks < NB / 4; ++ks) bf[ks] = Rk[(4 * ks + lk) * TSL + lr];
      auto tile = [&](int tcur, const float (&af)[2][NB / 4]) {
        const int j = TRANS ? tcur : kb + 1 + tcur;
        float* Rj = sY + (size_t)j * NB * TSL;
        gf32x4 c0v = gf32x4{0.f, 0.f, 0.f, 0.f}, c1v = gf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {        // the two row halves interleaved: independent accumulators back to back
          c0v = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0][ks], bf[ks], c0v, 0, 0, 0);
          c1v = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1][ks], bf[ks], c1v, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float* d0 = Rj + (4 * lk + r) * TSL + lr;
          float* d1 = Rj + (16 + 4 * lk + r) * TSL + lr;
          *d0 = *d0 - c0v[r];
          *d1 = *d1 - c1v[r];
        }
      };
      int t = wave;
      while (t < ntile) {                            // the next tile's operands travel while this one is multiplied
        if (t + TNW < ntile) loadA(kb, t + TNW, afB);
        tile(t, afA);
        t += TNW;
        if (t >= ntile) break;
        if (t + TNW < ntile) loadA(kb, t + TNW, afA);
        tile(t, afB);
        t += TNW;
      }
    }
    // first tile of the next step (its factor tile does not depend on this step's result)
    if (step + 1 < nbn && wave < (TRANS ? kbn : nbn - 1 - kbn)) loadA(kbn, wave, afA);
    __syncthreads();
  }
  // ---- results -------------------------------------------------------------------------------
  if (MODE == 0) {
    for (int c = 0; c < TSW && l0 + c < nd; ++c)
      for (int i = tid; i < np; i += NT) GP_VEC(vec_all, 1, l0 + c)[i] = i < n ? sY[i * TSL + c] : 0.f;
  } else if (MODE == 1) {
    if (slab >= nslab) {
      for (int c = 0; c < TSW && l0 + c < nd; ++c)
        for (int i = tid; i < np; i += NT) GP_VEC(vec_all, 2, l0 + c)[i] = i < n ? sY[i * TSL + c] : 0.f;
    } else {
      float* X = Out_all + (size_t)b * batch_stride;
      for (int e = tid; e < nrow * TSW; e += NT) {
        const int i = e / TSW, c = e % TSW;
        X[(size_t)i * np + c0 + c] = sY[i * TSL + c];
      }
    }
  } else {
    float* Sm = Out_all + (size_t)b * batch_stride;  // S[c0 + c][i] = Y[i][c]
    for (int e = tid; e < nrow * TSW; e += NT) {
      const int c = e / nrow, i = e - c * nrow;
      Sm[(size_t)(c0 + c) * np + i] = sY[i * TSL + c];
    }
  }
}

// r = u - v, v = row n of the factor (the forward-solved rhs); g_u = q -> g_Um; g_p = -a -> gp_rows  (what k_vec_a does after
// its product with the explicit inverse); one thread per element
// gpack_ind != nullptr (first half only): g_nu as well -- k_gnu's element of the same index, one launch less on the solve route
__global__ void k_vec_rv(int kernel, int Do, int n, int np, const float* __restrict__ Lall, size_t batch_stride,
                         const float* __restrict__ Dfac_all, size_t dfac_stride, const float* __restrict__ u,
                         float* __restrict__ vec_all, int with_a, float* __restrict__ gu_rows, float* __restrict__ gp_rows,
                         size_t u_dstride, const float* __restrict__ gpack_ind, const float* __restrict__ var, int Di,
                         size_t pack_dstride) {
  const int b = blockIdx.y, nb = gridDim.y, l = blockIdx.z, nd = gridDim.z, j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= np) return;
  if (gpack_ind) {
    float v = 0.f;
    if (j < n) {
      const int RQ2 = cdiv(Di + Do, 4);
      int m, d;
      if (kernel == 0) { m = j; d = b; } else { m = j / Do; d = j % Do; }
      const int field = Di + d;
      const float gc = (gpack_ind + (size_t)l * pack_dstride)[(((size_t)(m >> 6) * RQ2 + (field >> 2)) * 64 + (m & 63)) * 4 + (field & 3)];
      v = kernel == 0 ? gc * var[d] : gc;
    }
    GP_VEC(vec_all, 0, l)[j] = v;
  }
  const float* Lm = Lall + (size_t)b * batch_stride;
  const float* Dfac = Dfac_all + (size_t)b * dfac_stride;
  const int u_stride = kernel == 0 ? Do : 1, u_b = kernel == 0 ? b : 0;
  u += (size_t)l * u_dstride; gu_rows += (size_t)l * u_dstride; gp_rows += (size_t)l * u_dstride;
  if (!with_a) {
    float y = 0.f, rr = 0.f;
    if (j < n) {
      const int rn = n + l, kl = rn / NB, cl = kl * NB;        // draw l's forward-solved rhs is row n + l of the factor
      y = (j < cl) ? Lm[(size_t)rn * np + j] : Dfac[(size_t)kl * NB * NB + (rn - cl) * NB + (j - cl)];
      rr = u[(size_t)j * u_stride + u_b] - y;
    }
    GP_VEC(vec_all, 3, l)[j] = rr;
    GP_VEC(vec_all, 4, l)[j] = y;
  } else if (j < n) {
    gu_rows[(size_t)j * u_stride + u_b] = GP_VEC(vec_all, 1, l)[j];
    gp_rows[(size_t)j * u_stride + u_b] = -GP_VEC(vec_all, 2, l)[j];
  }
}

// ---------------------------------------------------------------------------------------------
// kernel-matrix backward.  G = (S + S^T)/2 multiplies BOTH triangles of K(Z) (torch's cholesky_backward).
// ---------------------------------------------------------------------------------------------
// RBF: grid (M, Do), block 64: one wavefront per (row point n, output dimension d); lanes stride over the other point m.
//   gZpart[d][n][i];  kpart[d][n]: [gvar | gell(Di)]  (k_chain_rbf sums them over n in a fixed order)
__global__ __launch_bounds__(64) void k_Kbwd_rbf(int Di, int Do, int M, int np, const float* __restrict__ Z,
                                                  const float* __restrict__ ell, const float* __restrict__ var,
                                                  const float* __restrict__ S_all, float* __restrict__ gZpart,
                                                  float* __restrict__ kpart) {
  const int n = blockIdx.x, d = blockIdx.y, lane = threadIdx.x;
  const float* Sm = S_all + (size_t)d * np * np;
  float gz[16], gl[16], zn[16], il[16];
  float gv = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    gz[i] = 0.f; gl[i] = 0.f;
    zn[i] = i < Di ? Z[n * Di + i] : 0.f;
    il[i] = i < Di ? 1.f / ell[d * Di + i] : 0.f;
  }
  const float vd = var[d];
  for (int m = lane; m < M; m += 64) {
    float qd = 0.f, dl[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      dl[i] = i < Di ? zn[i] - Z[m * Di + i] : 0.f;
      const float t = dl[i] * il[i];
      qd = fmaf(t, t, qd);
    }
    const float E = expf(-0.5f * qd);
    const float G = 0.5f * (Sm[(size_t)n * np + m] + Sm[(size_t)m * np + n]);
    const float GK = G * vd * E;
    gv = fmaf(G, E, gv);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      gz[i] = fmaf(-2.f * GK, dl[i] * il[i] * il[i], gz[i]);
      gl[i] = fmaf(GK, dl[i] * dl[i] * il[i] * il[i] * il[i], gl[i]);
    }
  }
  float* kp = kpart + ((size_t)d * M + n) * (1 + Di);
  gv = wave_allreduce_sum(gv);
  if (lane == 0) kp[0] = gv;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (i < Di) {                                    // wave-uniform
      const float sz = wave_allreduce_sum(gz[i]), sl = wave_allreduce_sum(gl[i]);
      if (lane == 0) {
        gZpart[((size_t)d * M + n) * Di + i] = sz;
        kp[1 + i] = sl;
      }
    }
  }
}

// DF: grid M, block 64 * NWK: point n per workgroup; wavefront w takes the rows a = w, w + NWK, ... of the D x D kernel block, its lanes
// stride over the other point m.  (One wavefront for all 36 entries of a block was a 2000-instruction body run twice by 36 lanes: 29 us on
// the tail of the backward pass's side branch at configs[1].)  The wavefronts' sums meet in LDS and are added in wavefront order.
//   kpart[n]: [gvar(D) | gell(D*D)],  gZpart[n][c]
template <int D> constexpr int kbwd_df_waves() { return D <= 8 ? D : 8; }
template <int N> __device__ __forceinline__ float kb_pick(const float (&v)[N], int idx) {   // v[idx], idx wave-uniform, no register indexing
  float r = v[0];
#pragma unroll
  for (int i = 1; i < N; ++i) r = (idx == i) ? v[i] : r;
  return r;
}
template <int D>
__global__ __launch_bounds__(64 * kbwd_df_waves<D>()) void k_Kbwd_df(int M, int np, const float* __restrict__ Z, const float* __restrict__ ell,
                                                                       const float* __restrict__ var, const float* __restrict__ Sm,
                                                                       float* __restrict__ gZpart, float* __restrict__ kpart) {
  constexpr int NWK = kbwd_df_waves<D>();
  constexpr int AR = (D + NWK - 1) / NWK;            // rows a per wavefront
  __shared__ float sred[NWK][2 * D];                 // per wavefront: gz[D] | gvar[D]
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float gz[D], gvar[D], gell[AR][D], zn[D];
#pragma unroll
  for (int a = 0; a < D; ++a) { gz[a] = 0.f; gvar[a] = 0.f; zn[a] = Z[n * D + a]; }
#pragma unroll
  for (int q = 0; q < AR; ++q)
#pragma unroll
    for (int b = 0; b < D; ++b) gell[q][b] = 0.f;
  for (int m = lane; m < M; m += 64) {
    float dl[D];
    float r2 = 0.f;
#pragma unroll
    for (int a = 0; a < D; ++a) { dl[a] = Z[m * D + a] - zn[a]; r2 = fmaf(dl[a], dl[a], r2); }  // delta = z_m - z_n
    float gd[D];
#pragma unroll
    for (int a = 0; a < D; ++a) gd[a] = 0.f;
#pragma unroll
    for (int q = 0; q < AR; ++q) {
      const int a = wv + NWK * q;                    // wave-uniform
      if (a < D) {
#pragma unroll
        for (int b = 0; b < D; ++b) {
          const float l = ell[a * D + b];
          const float il = 1.f / (l * l), il3 = il / l;  // 1/l^2, 1/l^3
          const float E = expf(-0.5f * r2 * il);
          const bool diag = a == b;
          const float da = kb_pick<D>(dl, a);
          const float term = da * dl[b] * il + (diag ? ((float)(D - 1) - r2 * il) : 0.f);
          const float vb = var[b];
          // entry (n,a),(m,b) and its point-transposed twin (m,a),(n,b): same value, same d/d z_n
          const size_t r1 = (size_t)(n * D + a), c1 = (size_t)(m * D + b);
          const size_t r2i = (size_t)(m * D + a), c2 = (size_t)(n * D + b);
          const float G1 = 0.5f * (Sm[r1 * np + c1] + Sm[c1 * np + r1]);
          const float G2 = 0.5f * (Sm[r2i * np + c2] + Sm[c2 * np + r2i]);
          const float base = vb * E * il;
          // dT/d delta_c = base * [ -il delta_c term + il (d_ca delta_b + d_cb delta_a) - diag 2 il delta_c ]
          const float Gs = (G1 + G2) * base;
          const float common = -il * term - (diag ? 2.f * il : 0.f);
#pragma unroll
          for (int c = 0; c < D; ++c) gd[c] = fmaf(Gs, common * dl[c] + (c == a ? il * dl[b] : 0.f), gd[c]);
          gd[b] = fmaf(Gs, il * da, gd[b]);
          // parameters: entry 1 only
          gvar[b] = fmaf(G1, E * il * term, gvar[b]);
          // dT/dl = vb E / l^3 [ r2 il term - 2 term - 2 il (delta_a delta_b - diag r2) ]
          const float dT = vb * E * il3 * (r2 * il * term - 2.f * term - 2.f * il * (da * dl[b] - (diag ? r2 : 0.f)));
          gell[q][b] = fmaf(G1, dT, gell[q][b]);
        }
      }
    }
#pragma unroll
    for (int c = 0; c < D; ++c) gz[c] -= gd[c];  // d/d z_n = - d/d delta
  }
  float red[D], tmp[D];
  wave_sum_all<D>(gz, red);
  if (lane == 0)
#pragma unroll
    for (int c = 0; c < D; ++c) sred[wv][c] = red[c];
  wave_sum_all<D>(gvar, red);
  if (lane == 0)
#pragma unroll
    for (int c = 0; c < D; ++c) sred[wv][D + c] = red[c];
  float* kp = kpart + (size_t)n * (D + D * D);
#pragma unroll
  for (int q = 0; q < AR; ++q) {
    const int a = wv + NWK * q;
#pragma unroll
    for (int b = 0; b < D; ++b) tmp[b] = gell[q][b];
    wave_sum_all<D>(tmp, red);
    if (lane == 0 && a < D)
#pragma unroll
      for (int b = 0; b < D; ++b) kp[D + a * D + b] = red[b];
  }
  __syncthreads();
  if (threadIdx.x < 2 * D) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NWK; ++w) v += sred[w][threadIdx.x];
    if (threadIdx.x < D) gZpart[(size_t)n * D + threadIdx.x] = v;
    else kp[threadIdx.x - D] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// final chains
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float rec_field(const float* __restrict__ base, size_t rec_f4_base, int lane, int field) {
  return base[((rec_f4_base + (size_t)(field >> 2)) * 64 + lane) * 4 + (field & 3)];
}

// RBF: one workgroup per output scalar (Do * Di lengthscales, Do variances), threads over the summation index, fixed-order
// reduction; the remaining workgroups assemble g_Z.  g_ell[d,i], g_var[d] -> raw.
__global__ __launch_bounds__(256) void k_chain_rbf(int Di, int Do, int M, int S, const float* __restrict__ pack_all,
                                                    const float* __restrict__ gpack_all, const float* __restrict__ raw_ell,
                                                    const float* __restrict__ raw_var, const float* __restrict__ nu_all,
                                                    const float* __restrict__ vjpZ_all, const float* __restrict__ gZpart,
                                                    const float* __restrict__ kpart,
                                                    float* __restrict__ g_raw_ell, float* __restrict__ g_raw_var,
                                                    float* __restrict__ g_Z, int nd, size_t pack_dstride) {
  // the pack chains of the nd draws (each with its own frequencies, weights and nu) are summed draw by draw in a fixed order
  const int RQ = cdiv(Di + 2, 4), RQ2 = cdiv(Di + Do, 4), SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  const size_t rff_f4 = (size_t)SJ * Do * RQ * 64, ind_f4 = (size_t)MJ * RQ2 * 64;
  const int tid = threadIdx.x, blk = blockIdx.x;
  __shared__ float red[4];
  auto block_sum = [&](float v) {                    // every thread gets the total
    v = wave_allreduce_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  if (blk < Do * Di) {
    const int e = blk, d = e / Di, i = e % Di;
    const float l = softplus_l(raw_ell[e]);
    float g = 0.f, gu = 0.f;
    for (int dr = 0; dr < nd; ++dr) {
      const float* pack = pack_all + (size_t)dr * pack_dstride;
      const float* gpack = gpack_all + (size_t)dr * pack_dstride;
      for (int s = tid; s < S; s += 256) {
        const size_t base = (size_t)((s >> 6) * Do + d) * RQ;
        g = fmaf(rec_field(gpack, base, s & 63, i), -rec_field(pack, base, s & 63, i) / l, g);  // om = eps/(2 pi l)
      }
      gu += gpack[4 * (rff_f4 + ind_f4) + e];
    }
    float k = 0.f;
    for (int n = tid; n < M; n += 256) k += kpart[((size_t)d * M + n) * (1 + Di) + 1 + i];
    g = block_sum(g);
    k = block_sum(k);
    if (tid == 0) {
      g = fmaf(gu, GP_LOG2E / (l * l * l), g);                                                // wl = -log2e/(2 l^2)
      g_raw_ell[e] = (g + k) * sigmoid_raw(raw_ell[e]);
    }
  } else if (blk < Do * Di + Do) {
    const int d = blk - Do * Di;
    const float v = softplus_l(raw_var[d]);
    float g = 0.f, c = 0.f, k = 0.f;
    for (int dr = 0; dr < nd; ++dr) {
      const float* pack = pack_all + (size_t)dr * pack_dstride;
      const float* gpack = gpack_all + (size_t)dr * pack_dstride;
      const float* gind = gpack + 4 * rff_f4;
      const float* nu = nu_all + (size_t)dr * Do * M;
      for (int s = tid; s < S; s += 256) {
        const size_t base = (size_t)((s >> 6) * Do + d) * RQ;
        g = fmaf(rec_field(gpack, base, s & 63, Di + 1), rec_field(pack, base, s & 63, Di + 1) / (2.f * v), g);  // aw = sqrt(v/S) w
      }
      for (int m = tid; m < M; m += 256)
        c = fmaf(rec_field(gind, (size_t)(m >> 6) * RQ2, m & 63, Di + d), nu[(size_t)d * M + m], c);             // cc = v nu
    }
    for (int m = tid; m < M; m += 256) k += kpart[((size_t)d * M + m) * (1 + Di)];
    g = block_sum(g);
    c = block_sum(c);
    k = block_sum(k);
    if (tid == 0) g_raw_var[d] = ((g + c) + k) * sigmoid_raw(raw_var[d]);
  } else {
    const int e = (blk - Do * Di - Do) * 256 + tid;
    if (e < M * Di) {
      const int m = e / Di, i = e % Di;
      float g = 0.f;
      for (int dr = 0; dr < nd; ++dr)
        g += rec_field(gpack_all + (size_t)dr * pack_dstride + 4 * rff_f4, (size_t)(m >> 6) * RQ2, m & 63, i) + vjpZ_all[(size_t)dr * M * Di + e];
      for (int d = 0; d < Do; ++d) g += gZpart[((size_t)d * M + m) * Di + i];
      g_Z[e] = g;
    }
  }
}

// DF step 1: per feature s, gradient w.r.t. omega[p,s,q] from the om fields and through B(omega):
//   gom[s][p*D+q]
// One workgroup per feature, thread (p, q): the D x D tables sit in LDS (one thread per feature with five D x D register arrays
// was 0.87 ms at D = 16 -- four 64-thread workgroups, spilling -- and 13 us on the critical tail of the step at D = 6).
template <int D>
__global__ __launch_bounds__(((D * D + 63) / 64) * 64) void k_df_gomega(int S, const float* __restrict__ pack, const float* __restrict__ gpack,
                                                                         const float* __restrict__ var, float* __restrict__ gom,
                                                                         size_t pack_dstride) {
  constexpr int RQ = cdiv(2 * D + 3, 4);
  __shared__ float om[D][D + 1], gB[D][D + 1], G[D][D + 1], nrm[D];   // om[p][q] = omega[p,s,q]
  pack += blockIdx.y * pack_dstride; gpack += blockIdx.y * pack_dstride; gom += (size_t)blockIdx.y * S * D * D;   // blockIdx.y = draw
  const int s = blockIdx.x, t = threadIdx.x;
  const int lane = s & 63, j0 = s >> 6;
  const bool on = t < D * D;
  const int p = on ? t / D : 0, q = on ? t % D : 0;
  float go = 0.f;
  if (on) {
    // thread (k = p, i = q) of record (s, i): fields om_k = omega[k,s,i]/(2 pi), bs_j
    const size_t base = (size_t)(j0 * D + q) * RQ;
    om[p][q] = rec_field(pack, base, lane, p) * GP_2PI;
    go = rec_field(gpack, base, lane, p) * GP_INV2PI;
    // gB[i][j]: thread (i = p, j = q)
    gB[p][q] = rec_field(gpack, (size_t)(j0 * D + p) * RQ, lane, D + 3 + q) * sqrtf(var[q] / (float)S);
  }
  __syncthreads();
  if (t < D) {
    float n2 = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) n2 = fmaf(om[k][t], om[k][t], n2);
    nrm[t] = sqrtf(n2);
  }
  if (on) {
    float g = 0.f;
#pragma unroll
    for (int k = 0; k < D; ++k) g = fmaf(om[p][k], om[q][k], g);
    G[p][q] = g;
  }
  __syncthreads();
  if (on) {
    // B[i][j] = nrm_j d_ij - G[i][j]/nrm_j
    float t1 = 0.f;
#pragma unroll
    for (int i = 0; i < D; ++i) t1 = fmaf(gB[i][q], ((i == q) ? 1.f : 0.f) + G[i][q] / (nrm[q] * nrm[q]), t1);
    float g = t1 * om[p][q] / nrm[q];
#pragma unroll
    for (int j = 0; j < D; ++j) g = fmaf(-gB[p][j] / nrm[j], om[j][q], g);
#pragma unroll
    for (int i = 0; i < D; ++i) g = fmaf(-gB[i][p] / nrm[p], om[i][q], g);
    gom[(size_t)s * D * D + p * D + q] = go + g;
  }
}

// DF step 2: chain rule to the raw parameters.  One workgroup per output scalar (D*D lengthscales, D variances),
// threads over the summation index, fixed-order reduction; the remaining workgroups assemble g_Z.
template <int D>
__global__ __launch_bounds__(256) void k_chain_df(int M, int S, const float* __restrict__ pack_all, const float* __restrict__ gpack_all,
                                                   const float* __restrict__ raw_ell, const float* __restrict__ raw_var,
                                                   const float* __restrict__ gom_all, const float* __restrict__ vjpZ_all,
                                                   const float* __restrict__ gZpart, const float* __restrict__ kpart,
                                                   float* __restrict__ g_raw_ell, float* __restrict__ g_raw_var,
                                                   float* __restrict__ g_Z, int nd, size_t pack_dstride) {
  constexpr int RQ = cdiv(2 * D + 3, 4), RQ2 = cdiv(2 * D, 4);
  const int SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  const size_t rff_f4 = (size_t)SJ * D * RQ * 64, ind_f4 = (size_t)MJ * RQ2 * 64;
  const int tid = threadIdx.x, blk = blockIdx.x;
  __shared__ float red[4];
  auto block_sum = [&](float v) {                    // every thread gets the total
    const float in1[1] = {v};
    float out1[1];
    wave_sum_multi<1>(in1, out1);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = out1[0];
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  };
  if (blk < D * D) {
    const int e = blk, a = e / D, b = e % D;         // ell[a][b]; omega[k,s,i] = eps/ell[i][k] -> entry (a,b) <- omega[b,s,a]
    const float l = softplus_l(raw_ell[e]);
    float g = 0.f, gu0 = 0.f, gu1 = 0.f;
    for (int dr = 0; dr < nd; ++dr) {                // the draws (own frequencies each) in a fixed order
      const float* pack = pack_all + (size_t)dr * pack_dstride;
      const float* guni = gpack_all + (size_t)dr * pack_dstride + 4 * (rff_f4 + ind_f4);
      const float* gom = gom_all + (size_t)dr * S * D * D;
      for (int s = tid; s < S; s += 256) {
        const size_t base = (size_t)((s >> 6) * D + a) * RQ;                // record (s, i=a), field k=b
        const float om = rec_field(pack, base, s & 63, b) * GP_2PI;         // omega[b,s,a]
        g = fmaf(gom[(size_t)s * D * D + b * D + a], -om / l, g);
      }
      gu0 += guni[e];
      gu1 += guni[D * D + e];
    }
    float k = 0.f;
    for (int n = tid; n < M; n += 256) k += kpart[(size_t)n * (D + D * D) + D + e];
    g = block_sum(g);
    k = block_sum(k);
    if (tid == 0) {
      const float l3 = l * l * l;
      g = fmaf(gu0, GP_LOG2E / l3, g);               // wab = -log2e/(2 l^2)
      g = fmaf(gu1, -2.f / l3, g);                   // il2 = 1/l^2
      g_raw_ell[e] = (g + k) * sigmoid_raw(raw_ell[e]);
    }
  } else if (blk < D * D + D) {
    const int j = blk - D * D;
    const float v = softplus_l(raw_var[j]);
    float g = 0.f, gu = 0.f;
    for (int dr = 0; dr < nd; ++dr) {
      const float* pack = pack_all + (size_t)dr * pack_dstride;
      const float* gpack = gpack_all + (size_t)dr * pack_dstride;
      for (int q = tid; q < S * D; q += 256) {
        const int s = q / D, i = q % D;
        const size_t base = (size_t)((s >> 6) * D + i) * RQ;
        g = fmaf(rec_field(gpack, base, s & 63, D + 3 + j), rec_field(pack, base, s & 63, D + 3 + j) / (2.f * v), g);  // bs = B sqrt(v/S)
      }
      gu += gpack[4 * (rff_f4 + ind_f4) + 2 * D * D + j];
    }
    float k = 0.f;
    for (int n = tid; n < M; n += 256) k += kpart[(size_t)n * (D + D * D) + j];
    g = block_sum(g);
    k = block_sum(k);
    if (tid == 0) g_raw_var[j] = (gu + g + k) * sigmoid_raw(raw_var[j]);
  } else {
    for (int e = (blk - D * D - D) * 256 + tid; e < M * D; e += (gridDim.x - D * D - D) * 256) {
      const int m = e / D, i = e % D;
      float g = 0.f;
      for (int dr = 0; dr < nd; ++dr)
        g += rec_field(gpack_all + (size_t)dr * pack_dstride + 4 * rff_f4, (size_t)(m >> 6) * RQ2, m & 63, i) + vjpZ_all[(size_t)dr * M * D + e];
      g_Z[e] = g + gZpart[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
// Which factors take the solve-based route: mode 0 "auto" (default): the LDS-resident sizes np <= 192 (BASELINE configs[0], [2],
// [3]: six 128-row systems, where the solves cost nothing measurable); mode 1 "always": every factor whose slab fits LDS
// (np <= 1216); mode 2 "never".  Past 192 rows the explicit inverse is faster (BASELINE configs[1], 608 rows: 3.25 vs 3.41 ms per
// training step) and as accurate while the factor is well conditioned; the training loop switches to "always" when the pivots of
// the factor approach the jitter floor (gpode_cache_pivots; main.py --backward_solves adaptive).
static int g_solves_mode = [] {
  const char* e = getenv("GPODE_BWD_SOLVES");
  if (e && e[0] == '1') return 1;
  const char* f = getenv("GPODE_BWD_EXPLICIT_INVERSE");
  return (f && f[0] == '1') ? 2 : 0;
}();
void set_backward_solves(int mode) { g_solves_mode = mode; }
int get_backward_solves() { return g_solves_mode; }
static inline bool use_trsm(int np) {
  if (g_solves_mode == 2) return false;
  return g_solves_mode == 1 ? np <= kTrsmMaxNp : np <= 192;
}

int cache_bwd_sizes(int kernel, int Di, int Do, int M, int S, int nd, size_t* bws_floats) {
  size_t pf = 0;
  if (cache_sizes(kernel, Di, Do, M, S, &pf, nullptr)) return 1;
  *bws_floats = bws_layout(kernel, Di, Do, M, S, pf, nd).total;
  return 0;
}

// The gradient-independent part of the cache backward: L^-1 from the factor the forward left in ws.  It needs nothing
// from the backward pass, so a caller may run it any time after the forward (e.g. on a side stream under the decoder)
// and pass prepared = 1 to cache_build_bwd with the same bws.  (nd only locates the factor in ws: it is shared by the draws.)
int cache_bwd_prepare(int kernel, int Di, int Do, int M, int S, int nd, const float* ws, float* bws, hipStream_t st) {
  size_t pf = 0;
  if (cache_sizes(kernel, Di, Do, M, S, &pf, nullptr)) return 1;
  const WsLayout w = ws_layout(kernel, Di, Do, M, S, nd);
  const BwsLayout b = bws_layout(kernel, Di, Do, M, S, pf, nd);
  const float* Lmat = ws + w.Lmat;
  const float* Dfac = ws + w.Dfac;
  const size_t bstride = (size_t)w.np * w.np, dstride = (size_t)w.nblk * NB * NB, dinv_stride = (size_t)b.nbn * NB * NB;
  hipLaunchKernelGGL(k_trinv_diag, dim3(b.nbn, b.batch), 64, 0, st, Dfac, dstride, b.n, bws + b.Dinv, dinv_stride);
  if (use_trsm(b.np)) return check_launch("cache bwd: diagonal-block inverses");   // the solve-based route needs nothing else
  hipLaunchKernelGGL(k_linv_init, dim3(b.nbn, b.nbn, b.batch), 256, 0, st, b.np, bws + b.Dinv, dinv_stride, bws + b.Linv, bstride);
  for (int sb = 1; sb < b.nbn; sb *= 2) {            // T lives in the X buffer (written by k_gemm_phiX only afterwards)
    const int npairs = cdiv(b.nbn, 2 * sb);
    if (sb >= GT / NB && big_factor(b.np)) {   // super-blocks of whole 128-row tiles: matrix cores
      const dim3 grid(sb / (GT / NB), sb / (GT / NB), npairs * b.batch);
      hipLaunchKernelGGL(k_linv_dc_mfma, grid, 256, 0, st, 0, sb, npairs, b.nbn, b.n, b.np, Lmat, bws + b.Linv, bws + b.X, bstride);
      hipLaunchKernelGGL(k_linv_dc_mfma, grid, 256, 0, st, 1, sb, npairs, b.nbn, b.n, b.np, Lmat, bws + b.Linv, bws + b.X, bstride);
      continue;
    }
    hipLaunchKernelGGL(k_linv_dc, dim3(sb, sb, npairs * b.batch), 256, 0, st, 0, sb, npairs, b.nbn, b.n, b.np, Lmat, bws + b.Linv, bws + b.X, bstride);
    hipLaunchKernelGGL(k_linv_dc, dim3(sb, sb, npairs * b.batch), 256, 0, st, 1, sb, npairs, b.nbn, b.n, b.np, Lmat, bws + b.Linv, bws + b.X, bstride);
  }
  return check_launch("cache bwd: L^-1");
}

// nd draws that shared one cache build: gpack, pack, eps_u are stacked along a leading draw axis; the parameter gradients
// come out summed over the draws (what autograd accumulates over the `for l in range(L)` loop of odegpvae.py:41-43).
int cache_build_bwd(int kernel, int Di, int Do, int M, int S, int nd, const float* raw_ell, const float* raw_var, const float* Z,
                    const float* eps_u, const float* pack, const float* ws, float* gpack, float* bws,
                    float* g_raw_ell, float* g_raw_var, float* g_Z, float* g_Um, float* g_Us, int prepared, hipStream_t st) {
  size_t pf = 0;
  if (cache_sizes(kernel, Di, Do, M, S, &pf, nullptr)) return 1;
  if (Di > 16 || Do > 16) return set_error("gpode_cache_build_bwd: D <= 16");
  const WsLayout w = ws_layout(kernel, Di, Do, M, S, nd);
  const BwsLayout b = bws_layout(kernel, Di, Do, M, S, pf, nd);
  const size_t SJ = cdiv(S, 64), MJ = cdiv(M, 64);
  const size_t rff_f4 = (kernel == 0 ? SJ * Do * cdiv(Di + 2, 4) : SJ * Do * cdiv(2 * Do + 3, 4)) * 64;
  const float* gpack_ind = gpack + 4 * rff_f4;
  const float* Lmat = ws + w.Lmat;
  const float* Dfac = ws + w.Dfac;
  const size_t bstride = (size_t)w.np * w.np, dstride = (size_t)w.nblk * NB * NB, dinv_stride = (size_t)b.nbn * NB * NB;
  const size_t MD = (size_t)M * Do;                  // per-draw stride of u / g_u / g_p rows
  float* vec = bws + b.vec;
  (void)MJ;

  const bool solves = use_trsm(b.np);
  if (!solves) hipLaunchKernelGGL(k_gnu, dim3(cdiv(b.np, 128), b.batch, nd), 128, 0, st, kernel, Di, Do, M, b.n, b.np, gpack_ind, ws + w.var, vec, pf);
  if (!(prepared & 1) && cache_bwd_prepare(kernel, Di, Do, M, S, nd, ws, bws, st)) return 1;
  const size_t trsm_lds = sizeof(float) * (size_t)(b.nbn + 2) * NB * TSL;
  const int nslab = (b.nbn * NB + TSW - 1) / TSW, vslab = cdiv(nd, TSW);     // slabs of Phi columns; slabs of per-draw vectors
  if (solves) {
    // triangular solves with the factor (torch's route); L^-1 is never formed
    if (set_max_lds((const void*)k_trsm_slab<0>, trsm_lds) || set_max_lds((const void*)k_trsm_slab<1>, trsm_lds) ||
        set_max_lds((const void*)k_trsm_slab<2>, trsm_lds)) return 1;
    hipLaunchKernelGGL(k_vec_rv, dim3(cdiv(b.np, 128), b.batch, nd), 128, 0, st, kernel, Do, b.n, b.np, Lmat, bstride, Dfac, dstride, ws + w.u, vec,
                       0, bws + b.gu_rows, bws + b.gp_rows, MD, gpack_ind, ws + w.var, Di, pf);
    hipLaunchKernelGGL(k_trsm_slab<0>, dim3(vslab, b.batch), 64 * TNW, trsm_lds, st, b.n, b.np, b.nbn, Lmat, bstride, Dfac, dstride, bws + b.Dinv, dinv_stride, vec,
                       (const float*)nullptr, (float*)nullptr, nd);
    hipLaunchKernelGGL(k_trsm_slab<1>, dim3(nslab + vslab, b.batch), 64 * TNW, trsm_lds, st, b.n, b.np, b.nbn, Lmat, bstride, Dfac, dstride, bws + b.Dinv, dinv_stride,
                       vec, (const float*)nullptr, bws + b.X, nd);
    // (g_u, g_p into (M, Do) rows: k_gUs_vec below)
  } else {
    hipLaunchKernelGGL(k_vec_q, dim3(cdiv(b.np, 4), b.batch, nd), 256, 0, st, b.n, b.np, bws + b.Linv, bstride, vec);
    hipLaunchKernelGGL(k_vec_a, dim3(cdiv(b.np, 4), b.batch, nd), 256, 0, st, kernel, Do, b.n, b.np, Lmat, bstride, Dfac, dstride, bws + b.Linv,
                       ws + w.u, vec, bws + b.gu_rows, bws + b.gp_rows, MD);
  }
  if (check_launch("cache bwd: solves")) return 1;
  {
    const size_t P = (size_t)M * (M + 1) / 2 * Do;   // >= M * Do: the same launch sums g_Um
    if (solves)
      hipLaunchKernelGGL(k_gUs_vec, (unsigned)((P + 255) / 256), 256, 0, st, kernel, M, Do, nd, b.batch, b.np, vec, eps_u, bws + b.gp_rows, g_Us, g_Um,
                         (prepared >> 1) & 1);
    else
      hipLaunchKernelGGL(k_gUs, (unsigned)((P + 255) / 256), 256, 0, st, M, Do, nd, bws + b.gu_rows, eps_u, g_Us, g_Um, (prepared >> 1) & 1);
  }
  // f_prior(Z) path, every draw through its own pack: d/dZ and parameter gradients (prior only), added to the pack gradient
  Draws dv; dv.nd = nd; dv.pack = pf; dv.in = 0; dv.in2 = MD; dv.out = (size_t)M * Di;
  if (rhs_vjp(kernel, Di, Do, M, S, pack, Z, bws + b.gp_rows, M, bws + b.vjpZ, 1, st, dv)) return 1;
  Draws dp; dp.nd = nd; dp.pack = pf; dp.in = 0; dp.in2 = MD; dp.out = pf;
  if (param_grad(kernel, Di, Do, M, S, pack, Z, bws + b.gp_rows, M, bws + b.slab, b.nchunkZ, gpack, 1, 1, st, dp)) return 1;
  // g_K = sym(L^-T Phi L^-1), Phi summed over the draws
  if (solves) {                      // X = L^-T Phi is there already; S^T = L^-T X^T (the consumer symmetrises S)
    hipLaunchKernelGGL(k_trsm_slab<2>, dim3(nslab, b.batch), 64 * TNW, trsm_lds, st, b.n, b.np, b.nbn, Lmat, bstride, Dfac, dstride, bws + b.Dinv, dinv_stride, vec,
                       bws + b.X, bws + b.S, nd);
  } else if (big_factor(b.np)) {     // big factor: both products on the matrix cores
    const int klim = b.nbn * NB, nt = cdiv(klim, GT);
    const dim3 grid(nt, nt, b.batch);                // column tile on the slow axis: the longest k ranges start first
    hipLaunchKernelGGL(k_gemm_mfma<1>, grid, 256, 0, st, bws + b.Linv, bws + b.Linv, bstride, b.np, klim, vec, bws + b.X, nd);
    hipLaunchKernelGGL(k_gemm_mfma<0>, grid, 256, 0, st, bws + b.X, bws + b.Linv, bstride, b.np, klim, vec, bws + b.S, nd);
  } else {
    hipLaunchKernelGGL(k_gemm_phiX, dim3(b.nbn, b.nbn, b.batch), 256, 0, st, bws + b.Linv, bstride, b.np, b.nbn, vec, bws + b.X, nd);
    hipLaunchKernelGGL(k_gemm_S, dim3(b.nbn, b.nbn, b.batch), 256, 0, st, bws + b.X, bws + b.Linv, bstride, b.np, b.nbn, bws + b.S);
  }
  if (check_launch("cache bwd: gK")) return 1;
  if (kernel == 0) {
    hipLaunchKernelGGL(k_Kbwd_rbf, dim3(M, Do), 64, 0, st, Di, Do, M, b.np, Z, ws + w.ell, ws + w.var, bws + b.S,
                       bws + b.gZpart, bws + b.kpart);
    hipLaunchKernelGGL(k_chain_rbf, Do * Di + Do + cdiv(M * Di, 256), 256, 0, st, Di, Do, M, S, pack, gpack, raw_ell, raw_var, ws + w.nu,
                       bws + b.vjpZ, bws + b.gZpart, bws + b.kpart, g_raw_ell, g_raw_var, g_Z, nd, pf);
    return check_launch("cache bwd: chain");
  }
#define X(D_)                                                                                                              \
  if (Do == D_) {                                                                                                          \
    hipLaunchKernelGGL(k_Kbwd_df<D_>, M, 64 * kbwd_df_waves<D_>(), 0, st, M, b.np, Z, ws + w.ell, ws + w.var, bws + b.S, bws + b.gZpart, \
                       bws + b.kpart);                                                                                     \
    hipLaunchKernelGGL(k_df_gomega<D_>, dim3(S, nd), ((D_ * D_ + 63) / 64) * 64, 0, st, S, pack, gpack, ws + w.var, bws + b.gom, pf);    \
    hipLaunchKernelGGL(k_chain_df<D_>, D_ * D_ + D_ + cdiv(M * D_, 256), 256, 0, st, M, S, pack, gpack, raw_ell, raw_var, bws + b.gom, bws + b.vjpZ,      \
                       bws + b.gZpart, bws + b.kpart, g_raw_ell, g_raw_var, g_Z, nd, pf);                                  \
    return check_launch("cache bwd: chain df");                                                                            \
  }
  X(6) X(4) X(2) X(3) X(8) X(16) X(5) X(7) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#undef X
  return set_error("gpode_cache_build_bwd: DF backward is built for D = 2 .. 16");
}

}  // namespace gp

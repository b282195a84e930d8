// vae_conv.hip -- conv VAE building blocks (encoder vae.py:47-98, decoder vae.py:101-153), NCHW fp32.
//
// Three convolution kernels cover all seven layers, forward and backward:
//   conv_fwd        y  = conv(x, w) + b                          Conv2d forward;  ConvTranspose2d d/d input
//   conv_bwd_data   gx = sum_co sum_taps gy * w  (+ b)           Conv2d d/d input; ConvTranspose2d FORWARD
//   conv_bwd_weight gw = sum_{b,oy,ox} gy * x                    both (roles of x / gy swap for ConvTranspose2d)
// nn.ConvTranspose2d stores its weight as (C_in, C_out, k, k), which is exactly the (Co, Ci, k, k) weight of
// the convolution it is the adjoint of -- the same buffer serves both directions.
// K (kernel size) and STRIDE are template parameters so the tap loops unroll and the stride test of the
// transposed direction folds into the loop bounds (only contributing taps are visited).
//
// Round-1 status: direct (gather) form, one thread per output element, weights/inputs through L1/L2.
// This is the correctness baseline of the decoder; the LDS-resident implicit-GEMM MFMA version of the two
// FLOP-dominant layers (decnn.4, decnn.7) is the next optimisation step.
#include <hip/hip_runtime.h>
#include "gp_launch.hpp"
#include "bn_sink.hpp"
#include "wave_reduce.hpp"

namespace gp {

struct ConvDims { int B, Ci, H, W, Co, P, Ho, Wo; size_t xbs; };   // xbs: floats between consecutive images of x (dense: Ci * H * W)
int chan_sum(const float* v, float* out, int B, int C, int HW, float* scratch, hipStream_t st);

template <int K, int S>
__global__ __launch_bounds__(256) void k_conv_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ bias, float* __restrict__ y, ConvDims d) {
  const size_t total = (size_t)d.B * d.Co * d.Ho * d.Wo;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(e % d.Wo), oy = (int)((e / d.Wo) % d.Ho), co = (int)((e / ((size_t)d.Wo * d.Ho)) % d.Co);
    const int b = (int)(e / ((size_t)d.Wo * d.Ho * d.Co));
    float acc = bias ? bias[co] : 0.f;
    const int iy0 = oy * S - d.P, ix0 = ox * S - d.P;
    const float* xb = x + (size_t)b * d.xbs;
    const float* wc = w + (size_t)co * d.Ci * K * K;
    for (int ci = 0; ci < d.Ci; ++ci) {
      const float* xc = xb + (size_t)ci * d.H * d.W;
      const float* wk = wc + ci * K * K;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int iy = iy0 + ky;
        if (iy < 0 || iy >= d.H) continue;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int ix = ix0 + kx;
          if (ix < 0 || ix >= d.W) continue;
          acc = fmaf(xc[iy * d.W + ix], wk[ky * K + kx], acc);
        }
      }
    }
    y[e] = acc;
  }
}

// gx[b,ci,iy,ix] = bias[ci] + sum_{co,ky,kx} gy[b,co,oy,ox] w[co,ci,ky,kx],  oy = (iy + P - ky)/S when divisible
template <int K, int S>
__global__ __launch_bounds__(256) void k_conv_bwd_data(const float* __restrict__ gy, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ gx, ConvDims d) {
  const size_t total = (size_t)d.B * d.Ci * d.H * d.W;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int ix = (int)(e % d.W), iy = (int)((e / d.W) % d.H), ci = (int)((e / ((size_t)d.W * d.H)) % d.Ci);
    const int b = (int)(e / ((size_t)d.W * d.H * d.Ci));
    float acc = bias ? bias[ci] : 0.f;
    const float* gb = gy + (size_t)b * d.Co * d.Ho * d.Wo;
    const int ky0 = (iy + d.P) % S, kx0 = (ix + d.P) % S;  // only taps with (iy + P - ky) % S == 0 contribute
    for (int co = 0; co < d.Co; ++co) {
      const float* gc = gb + (size_t)co * d.Ho * d.Wo;
      const float* wk = w + ((size_t)co * d.Ci + ci) * K * K;
#pragma unroll
      for (int t = 0; t < (K + S - 1) / S; ++t) {
        const int ky = ky0 + t * S;
        if (ky >= K) continue;
        const int oy = (iy + d.P - ky) / S;
        if (iy + d.P - ky < 0 || oy >= d.Ho) continue;
#pragma unroll
        for (int u = 0; u < (K + S - 1) / S; ++u) {
          const int kx = kx0 + u * S;
          if (kx >= K) continue;
          const int ox = (ix + d.P - kx) / S;
          if (ix + d.P - kx < 0 || ox >= d.Wo) continue;
          acc = fmaf(gc[oy * d.Wo + ox], wk[ky * K + kx], acc);
        }
      }
    }
    gx[e] = acc;
  }
}

// gw[co,ci,ky,kx] = sum_{b,oy,ox} gy[b,co,oy,ox] x[b,ci,oy*S-P+ky,ox*S-P+kx].  grid (Co*Ci, nsplit): each workgroup
// sums a slice of the batch for one (co,ci) pair into part[split][co][ci][K*K]; a second kernel adds the splits.
template <int K, int S>
__global__ __launch_bounds__(256) void k_conv_bwd_weight(const float* __restrict__ x, const float* __restrict__ gy,
                                                          float* __restrict__ part, ConvDims d, int b_per_split) {
  __shared__ float red[4][K * K];
  const int co = blockIdx.x / d.Ci, ci = blockIdx.x % d.Ci;
  const int b0 = blockIdx.y * b_per_split, b1 = min(d.B, b0 + b_per_split);
  float acc[K * K];
#pragma unroll
  for (int t = 0; t < K * K; ++t) acc[t] = 0.f;
  const int npix = d.Ho * d.Wo;
  const size_t work = (size_t)(b1 - b0) * npix;
  for (size_t e = threadIdx.x; e < work; e += 256) {
    const int b = b0 + (int)(e / npix), p = (int)(e % npix);
    const int oy = p / d.Wo, ox = p % d.Wo;
    const float g = gy[((size_t)b * d.Co + co) * npix + p];
    const float* xc = x + (size_t)b * d.xbs + (size_t)ci * d.H * d.W;
    const int iy0 = oy * S - d.P, ix0 = ox * S - d.P;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int iy = iy0 + ky;
      const bool oky = iy >= 0 && iy < d.H;
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const int ix = ix0 + kx;
        const float xv = (oky && ix >= 0 && ix < d.W) ? xc[iy * d.W + ix] : 0.f;
        acc[ky * K + kx] = fmaf(g, xv, acc[ky * K + kx]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < K * K; ++t) {
    const float s = row_allreduce16(acc[t]);  // 16-lane rows; the four rows are added below
    float tot = GP_LANE(s, 0) + GP_LANE(s, 16) + GP_LANE(s, 32) + GP_LANE(s, 48);
    if (lane == 0) red[wv][t] = tot;
  }
  __syncthreads();
  if (threadIdx.x < K * K)
    part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (K * K) + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void k_sum_splits(const float* __restrict__ part, int nsplit, size_t n, float* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float acc = 0.f;
  for (int s = 0; s < nsplit; ++s) acc += part[(size_t)s * n + e];
  out[e] = acc;
}

// elementwise activations: mode 0 relu, 1 sigmoid
__global__ void k_act_fwd(const float* __restrict__ x, float* __restrict__ y, size_t n, int mode) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float v = x[e];
    y[e] = mode == 0 ? fmaxf(v, 0.f) : 1.f / (1.f + expf(-v));
  }
}
__global__ void k_act_bwd(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ gx, size_t n, int mode) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float v = y[e];
    gx[e] = mode == 0 ? (v > 0.f ? gy[e] : 0.f) : gy[e] * v * (1.f - v);
  }
}

// Linear: y[b,o] = bias[o] + sum_i x[b,i] w[o,i]
__global__ void k_linear_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                             float* __restrict__ y, int B, int In, int Out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)B * Out) return;
  const int b = (int)(e / Out), o = (int)(e % Out);
  float acc = bias ? bias[o] : 0.f;
  const float* xr = x + (size_t)b * In;
  const float* wr = w + (size_t)o * In;
  for (int i = 0; i < In; ++i) acc = fmaf(xr[i], wr[i], acc);
  y[e] = acc;
}
// xpre != nullptr: the layer's input was relu(xpre) (applied on load by the forward) -- the gradient passes where xpre > 0
__global__ void k_linear_bwd_x(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx, int B, int In, int Out,
                               const float* __restrict__ xpre) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (size_t)B * In) return;
  const int b = (int)(e / In), i = (int)(e % In);
  float acc = 0.f;
  for (int o = 0; o < Out; ++o) acc = fmaf(gy[(size_t)b * Out + o], w[(size_t)o * In + i], acc);
  gx[e] = (xpre && !(xpre[e] > 0.f)) ? 0.f : acc;
}
// gw[o,i] = sum_b gy[b,o] x[b,i]; gb[o] = sum_b gy[b,o].  One wave per (o,i) (i == In -> bias).
__global__ __launch_bounds__(256) void k_linear_bwd_w(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ gw,
                                                       float* __restrict__ gb, int B, int In, int Out, int relu_in) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= Out * (In + 1)) return;
  const int o = wave / (In + 1), i = wave % (In + 1);
  float acc = 0.f;
  for (int b = lane; b < B; b += 64) {
    float xv = i < In ? x[(size_t)b * In + i] : 1.f;
    if (relu_in && i < In) xv = fmaxf(xv, 0.f);
    acc = fmaf(gy[(size_t)b * Out + o], xv, acc);
  }
  float in1[1] = {acc}, out1[1];
  wave_sum_multi<1>(in1, out1);
  if (lane == 0) { if (i < In) gw[(size_t)o * In + i] = out1[0]; else if (gb) gb[o] = out1[0]; }
}

// The same sums for the decoder's fc layer at thousands of rows (q -> 512 on batch x T latent states, vae.py:101): a workgroup owns 64
// consecutive outputs and a slab of rows -- gy is read in 256-byte rows (one wavefront per (o, i) pair re-read every cache line of gy
// 16 times through 2 KB strides: 36 us at 4096 rows), x[b][i] is wave-uniform.  partW[slab][o][i], partB[slab][o] are summed in a
// fixed order by reduce_jobs.  grid (Out / 64, nslab), block 256 = 64 outputs x 4 row groups.
constexpr int LINW_SLABS = 32;
__global__ __launch_bounds__(256) void k_linear_bwd_w_cols(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ partW,
                                                            float* __restrict__ partB, int B, int In, int Out, int rps) {
  __shared__ float red[4][9][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6, o = blockIdx.x * 64 + lane;
  const int b0 = blockIdx.y * rps, b1 = min(B, b0 + rps);
  float acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = 0.f;
#pragma unroll 4
  for (int b = b0 + g; b < b1; b += 4) {
    const float gv = gy[(size_t)b * Out + o];
    const float* xr = x + (size_t)b * In;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < In) acc[i] = fmaf(gv, xr[i], acc[i]);
    acc[8] += gv;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) red[g][i][lane] = acc[i];
  __syncthreads();
  for (int e = threadIdx.x; e < 9 * 64; e += 256) {
    const int i = e >> 6, l = e & 63, oo = blockIdx.x * 64 + l;
    const float v = (red[0][i][l] + red[1][i][l]) + (red[2][i][l] + red[3][i][l]);
    if (i < In) partW[((size_t)blockIdx.y * Out + oo) * In + i] = v;
    else if (i == 8) partB[(size_t)blockIdx.y * Out + oo] = v;
  }
}

// ---- wide fan-in, few outputs (the encoder's fc layer, 512 -> 2q on the minibatch, vae.py:62): one wavefront per output
//      element, lanes along the reduction
__global__ __launch_bounds__(256) void k_linear_fwd_fanin(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int B, int In, int Out,
                                                           int relu_in) {
  const int lane = threadIdx.x & 63;
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= B * Out) return;
  const int b = e / Out, o = e % Out;
  const float* xr = x + (size_t)b * In;
  const float* wr = w + (size_t)o * In;
  float acc = 0.f;
  if (relu_in) { for (int i = lane; i < In; i += 64) acc = fmaf(fmaxf(xr[i], 0.f), wr[i], acc); }
  else { for (int i = lane; i < In; i += 64) acc = fmaf(xr[i], wr[i], acc); }
  const float in1[1] = {acc};
  float out1[1];
  wave_sum_multi<1>(in1, out1);
  if (lane == 0) y[e] = out1[0] + (bias ? bias[o] : 0.f);
}

// ---- small fan-in (In <= 16, Out a multiple of 64): the decoder's fc layer (latent -> 512, vae.py:66) on all
//      batch*T latent states.  Threads run along Out (coalesced rows of y / gy); the few weights per output live in
//      registers.
constexpr int LIN_MAXIN = 16;
__global__ __launch_bounds__(256) void k_linear_fwd_fanout(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ y, int B, int In, int Out,
                                                            int rows_per_block) {
  const int b0 = blockIdx.x * rows_per_block, b1 = min(B, b0 + rows_per_block);
  // (grid.y splits Out: one pass per thread -- a second pass waited out a second round of weight loads)
  for (int o = blockIdx.y * 256 + threadIdx.x; o < Out; o += 256 * gridDim.y) {
    float wr[LIN_MAXIN];
#pragma unroll
    for (int i = 0; i < LIN_MAXIN; ++i) wr[i] = i < In ? w[(size_t)o * In + i] : 0.f;
    const float bo = bias ? bias[o] : 0.f;
    for (int b = b0; b < b1; ++b) {
      const float* xr = x + (size_t)b * In;          // wave-uniform
      float acc = bo;
#pragma unroll
      for (int i = 0; i < LIN_MAXIN; ++i)
        if (i < In) acc = fmaf(xr[i], wr[i], acc);
      y[(size_t)b * Out + o] = acc;
    }
  }
}
// gx[b][i] = sum_o gy[b][o] w[o][i]: one wavefront per row b, lanes along o
template <int OPL>  // outputs per lane = Out / 64
__global__ __launch_bounds__(256) void k_linear_bwd_x_fanout(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx,
                                                              int B, int In) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = gridDim.x * 4;
  constexpr int Out = OPL * 64;
  float wr[OPL][LIN_MAXIN];
#pragma unroll
  for (int k = 0; k < OPL; ++k)
#pragma unroll
    for (int i = 0; i < LIN_MAXIN; ++i) wr[k][i] = i < In ? w[(size_t)(lane + 64 * k) * In + i] : 0.f;
  for (int b = blockIdx.x * 4 + wave; b < B; b += nw) {
    float acc[LIN_MAXIN];
#pragma unroll
    for (int i = 0; i < LIN_MAXIN; ++i) acc[i] = 0.f;
#pragma unroll
    for (int k = 0; k < OPL; ++k) {
      const float g = gy[(size_t)b * Out + lane + 64 * k];
#pragma unroll
      for (int i = 0; i < LIN_MAXIN; ++i) acc[i] = fmaf(g, wr[k][i], acc[i]);
    }
    float tot[LIN_MAXIN];
    wave_sum_all<LIN_MAXIN>(acc, tot);
    if (lane < In) {
      float v = 0.f;
#pragma unroll
      for (int i = 0; i < LIN_MAXIN; ++i) v = (lane == i) ? tot[i] : v;
      gx[(size_t)b * In + lane] = v;
    }
  }
}
// Bernoulli log-likelihood (vae.py:136-153, no epsilon -- SURVEY F9): ll = log(z) X + log(1-z) (1-X).
// z has `reps` copies of X's extent (X.repeat([L,...])): X index = e % nX.
__global__ void k_loglik_fwd(const float* __restrict__ X, const float* __restrict__ z, float* __restrict__ ll, size_t n, size_t nX) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float xv = X[e % nX], zv = z[e];
    ll[e] = logf(zv) * xv + logf(1.f - zv) * (1.f - xv);
  }
}
__global__ void k_loglik_bwd(const float* __restrict__ X, const float* __restrict__ z, const float* __restrict__ g,
                             float* __restrict__ gz, size_t n, size_t nX) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float xv = X[e % nX], zv = z[e];
    gz[e] = g[e] * (xv / zv - (1.f - xv) / (1.f - zv));
  }
}
// fused reduction used by the ELBO (create_model.py:52-53): out[row] = sum over `inner` of ll; row = (l,n)
__global__ __launch_bounds__(256) void k_loglik_rowsum(const float* __restrict__ X, const float* __restrict__ z, float* __restrict__ out,
                                                        size_t inner, size_t nX) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  float acc = 0.f;
  for (size_t p = threadIdx.x; p < inner; p += 256) {
    const size_t e = row * inner + p;
    const float xv = X[e % nX], zv = z[e];
    acc += logf(zv) * xv + logf(1.f - zv) * (1.f - xv);
  }
  float in1[1] = {acc}, out1[1];
  wave_sum_multi<1>(in1, out1);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = out1[0];
  __syncthreads();
  if (threadIdx.x == 0) out[row] = red[0] + red[1] + red[2] + red[3];
}
__global__ void k_loglik_rowsum_bwd(const float* __restrict__ X, const float* __restrict__ z, const float* __restrict__ grow,
                                    float* __restrict__ gz, size_t n, size_t inner, size_t nX) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float xv = X[e % nX], zv = z[e];
    gz[e] = grow[e / inner] * (xv / zv - (1.f - xv) / (1.f - zv));
  }
}

// ---------------------------------------------------------------------------------------------
// host
// ---------------------------------------------------------------------------------------------
static inline int ew_grid(size_t n) { size_t g = (n + 255) / 256; return (int)(g < 8192 ? (g ? g : 1) : 8192); }

#define GP_CONV_KS(X) X(5, 2) X(5, 1) X(3, 1) X(3, 2)

// LDS-tiled specialisations of the heavy decoder layers (vae_conv_tiled.hip); -1 = geometry not covered
int tiled_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co, int K, int S, int P,
              int Ho, int Wo, hipStream_t st);
int tiled_bwd_data(const float* gy, const float* w, const float* bias, float* gx, int B, int Ci, int H, int W, int Co, int K, int S,
                   int P, int Ho, int Wo, const float* in_bn, hipStream_t st, const BnSink* sink);
int tiled_bwd_weight(const float* x, const float* gy, float* gw, float* scratch, int B, int Ci, int H, int W, int Co, int K, int S,
                     int P, int Ho, int Wo, const float* in_bn, hipStream_t st);

// xbs != 0: x is batch-strided (xbs floats between images, each image dense) -- generic kernels only
int conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co, int K, int S,
               int P, int Ho, int Wo, hipStream_t st, size_t xbs) {
  if (xbs == (size_t)Ci * H * W) xbs = 0;
  if (!xbs) { const int r = tiled_fwd(x, w, bias, y, B, Ci, H, W, Co, K, S, P, Ho, Wo, st); if (r >= 0) return r; }
  else if (Ci % 4 == 0) return set_error("gpode_conv2d_fwd_bs: batch-strided input is for the generic kernels (input channels not a multiple of 4)");
  ConvDims d{B, Ci, H, W, Co, P, Ho, Wo, xbs ? xbs : (size_t)Ci * H * W};
  const size_t total = (size_t)B * Co * Ho * Wo;
#define X(k, s) if (K == k && S == s) { hipLaunchKernelGGL((k_conv_fwd<k, s>), ew_grid(total), 256, 0, st, x, w, bias, y, d); return check_launch("conv_fwd"); }
  GP_CONV_KS(X)
#undef X
  return set_error("gpode_conv2d_fwd: kernel %d stride %d not built", K, S);
}

// gy_bn (optional, [Co][4] = mean, invstd, gamma, beta): gy is the raw output of the previous layer and the BatchNorm + ReLU
// between the two layers is applied on the fly (matrix-core specialisations only)
// ConvTranspose2d forward that also produces the training-mode BatchNorm statistics of its output (bn_sink.hpp): matrix-core
// specialisations only.  scratch: convT_fwd_stats_scratch() floats; slot: which of the library's ticket counters this layer uses
// (launches that may run concurrently need different slots).
__device__ unsigned g_bn_sink_tickets[64];
size_t convT_fwd_stats_scratch(int Co_out) { return (size_t)512 * Co_out * 2; }
int convT_fwd_stats(const float* x, const float* x_bn, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co, int K,
                    int S, int P, int Ho, int Wo, const float* gamma, const float* beta, float* save_mean, float* save_invstd,
                    float* running_mean, float* running_var, long long* nbt, float momentum, float eps, float* table, float* scratch,
                    int slot, hipStream_t st) {
  static unsigned* tickets = nullptr;
  if (!tickets) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_bn_sink_tickets)) != hipSuccess) return set_error("gpode_convT_fwd_stats: no ticket storage");
    tickets = static_cast<unsigned*>(p);
  }
  if (slot < 0 || slot >= 64) return set_error("gpode_convT_fwd_stats: slot 0 .. 63");
  BnSink sink{scratch, tickets + slot, gamma, beta, save_mean, save_invstd, running_mean, running_var, nbt, table, momentum, eps,
              (float)B * (float)(H * W)};
  const int r = tiled_bwd_data(x, w, bias, y, B, Ci, H, W, Co, K, S, P, Ho, Wo, x_bn, st, &sink);
  return r < 0 ? set_error("gpode_convT_fwd_stats: no specialisation for this geometry") : r;
}

int conv2d_bwd_data(const float* gy, const float* w, const float* bias, float* gx, int B, int Ci, int H, int W, int Co, int K, int S,
                    int P, int Ho, int Wo, const float* gy_bn, hipStream_t st) {
  { const int r = tiled_bwd_data(gy, w, bias, gx, B, Ci, H, W, Co, K, S, P, Ho, Wo, gy_bn, st, nullptr); if (r >= 0) return r; }
  if (gy_bn) return set_error("gpode_conv2d_bwd_data_bn: no matrix-core specialisation for this geometry");
  ConvDims d{B, Ci, H, W, Co, P, Ho, Wo, (size_t)Ci * H * W};
  const size_t total = (size_t)B * Ci * H * W;
#define X(k, s) if (K == k && S == s) { hipLaunchKernelGGL((k_conv_bwd_data<k, s>), ew_grid(total), 256, 0, st, gy, w, bias, gx, d); return check_launch("conv_bwd_data"); }
  GP_CONV_KS(X)
#undef X
  return set_error("gpode_conv2d_bwd_data: kernel %d stride %d not built", K, S);
}

static inline int pick_split(int B, int nblocks_per_split) {
  int s = 1024 / (nblocks_per_split > 0 ? nblocks_per_split : 1);  // aim for ~1024 workgroups
  if (s < 1) s = 1;
  if (s > B) s = B;
  if (s > 64) s = 64;
  return s;
}

// scratch: conv_wgrad_scratch_floats(...) floats
size_t conv_wgrad_scratch(int B, int Ci, int Co, int K) {
  const size_t generic = (size_t)pick_split(B, Co * Ci) * Co * Ci * K * K + (size_t)64 * Co * 2;
  // the tiled weight-gradient kernels split the batch into up to 512 slabs (two workgroups per CU) of Co*Ci*K*K floats
  const size_t tiled = (size_t)(B < 512 ? B : 512) * Co * Ci * K * K + (size_t)64 * Co * 2;
  return generic > tiled ? generic : tiled;
}

int conv2d_bwd_weight(const float* x, const float* gy, float* gw, float* gbias, float* scratch, int B, int Ci, int H, int W, int Co,
                      int K, int S, int P, int Ho, int Wo, const float* gy_bn, hipStream_t st, size_t xbs) {
  if (xbs == (size_t)Ci * H * W) xbs = 0;
  if (xbs && Ci % 4 == 0) return set_error("gpode_conv2d_bwd_weight_bs: batch-strided input is for the generic kernels");
  if (!xbs) {
    const int r = tiled_bwd_weight(x, gy, gw, scratch, B, Ci, H, W, Co, K, S, P, Ho, Wo, gy_bn, st);
    if (r > 0) return r;
    if (r == 0) {
      // the bias partials go BEHIND the weight-gradient slabs (the tail conv_wgrad_scratch reserves): with deferred reductions
      // (gpode_defer_reductions) the slabs are still unread when k_chan_sum runs
      if (gbias) return chan_sum(gy, gbias, B, Co, Ho * Wo, scratch + (size_t)(B < 512 ? B : 512) * Co * Ci * K * K, st);
      return 0;
    }
  }
  if (gy_bn) return set_error("gpode_conv2d_bwd_weight_bn: no matrix-core specialisation for this geometry");
  ConvDims d{B, Ci, H, W, Co, P, Ho, Wo, xbs ? xbs : (size_t)Ci * H * W};
  const int nsplit = pick_split(B, Co * Ci);
  const int bps = (B + nsplit - 1) / nsplit;
  const int used = (B + bps - 1) / bps;
  const size_t n = (size_t)Co * Ci * K * K;
  bool ok = false;
#define X(k, s) if (K == k && S == s) { hipLaunchKernelGGL((k_conv_bwd_weight<k, s>), dim3(Co * Ci, used), 256, 0, st, x, gy, scratch, d, bps); ok = true; }
  GP_CONV_KS(X)
#undef X
  if (!ok) return set_error("gpode_conv2d_bwd_weight: kernel %d stride %d not built", K, S);
  if (reduce_job(RedJob{scratch, gw, used, (int)n, 0, 0, 0, 0}, st)) return 1;
  if (gbias) return chan_sum(gy, gbias, B, Co, Ho * Wo, scratch + (size_t)nsplit * n, st);
  return check_launch("conv_bwd_weight");
}

int act_fwd(const float* x, float* y, size_t n, int mode, hipStream_t st) {
  hipLaunchKernelGGL(k_act_fwd, ew_grid(n), 256, 0, st, x, y, n, mode);
  return check_launch("act_fwd");
}
int act_bwd(const float* y, const float* gy, float* gx, size_t n, int mode, hipStream_t st) {
  hipLaunchKernelGGL(k_act_bwd, ew_grid(n), 256, 0, st, y, gy, gx, n, mode);
  return check_launch("act_bwd");
}

int linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, hipStream_t st) {
  if (In <= LIN_MAXIN && Out % 64 == 0 && B >= 256) {
    const int rows = 8;
    hipLaunchKernelGGL(k_linear_fwd_fanout, dim3((B + rows - 1) / rows, (Out + 255) / 256), 256, 0, st, x, w, bias, y, B, In, Out, rows);
    return check_launch("linear_fwd_fanout");
  }
  if (In >= 128 && (size_t)B * Out <= (size_t)1 << 22) {
    hipLaunchKernelGGL(k_linear_fwd_fanin, (unsigned)(((size_t)B * Out + 3) / 4), 256, 0, st, x, w, bias, y, B, In, Out, 0);
    return check_launch("linear_fwd_fanin");
  }
  hipLaunchKernelGGL(k_linear_fwd, (unsigned)(((size_t)B * Out + 255) / 256), 256, 0, st, x, w, bias, y, B, In, Out);
  return check_launch("linear_fwd");
}
size_t linear_bwd_scratch(int B, int In, int Out) { (void)B; return (size_t)LINW_SLABS * Out * (In + 1); }

int linear_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int B, int In, int Out, float* scratch,
               hipStream_t st) {
  if (In <= LIN_MAXIN && Out % 64 == 0 && Out <= 512 && B >= 256) {
    const int nb = B / 4 < 256 ? (B + 3) / 4 : 256;
    if (gx) {
      switch (Out / 64) {
#define X(k) case k: hipLaunchKernelGGL(k_linear_bwd_x_fanout<k>, nb, 256, 0, st, gy, w, gx, B, In); break;
        X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8)
#undef X
      }
    }
    if (gw) {
      if (scratch && In <= 8 && B >= 1024) {
        const int rps = (B + LINW_SLABS - 1) / LINW_SLABS, used = (B + rps - 1) / rps;
        float* partW = scratch;
        float* partB = scratch + (size_t)LINW_SLABS * Out * In;
        hipLaunchKernelGGL(k_linear_bwd_w_cols, dim3(Out / 64, used), 256, 0, st, x, gy, partW, partB, B, In, Out, rps);
        const RedJob jobs[2] = {RedJob{partW, gw, used, Out * In, 0, 0, 0, 0}, RedJob{partB, gb, used, Out, 0, 0, 0, 0}};
        if (reduce_jobs(jobs, gb ? 2 : 1, st)) return 1;
      } else {
        hipLaunchKernelGGL(k_linear_bwd_w, (unsigned)(((size_t)Out * (In + 1) * 64 + 255) / 256), 256, 0, st, x, gy, gw, gb, B, In, Out, 0);
      }
    }
    return check_launch("linear_bwd_fanout");
  }
  if (gx) hipLaunchKernelGGL(k_linear_bwd_x, (unsigned)(((size_t)B * In + 255) / 256), 256, 0, st, gy, w, gx, B, In, Out, (const float*)nullptr);
  if (gw) hipLaunchKernelGGL(k_linear_bwd_w, (unsigned)(((size_t)Out * (In + 1) * 64 + 255) / 256), 256, 0, st, x, gy, gw, gb, B, In, Out, 0);
  return check_launch("linear_bwd");
}

// y = relu(x) W^T + b and its backward with the ReLU of the INPUT folded in (the encoder's cnn.6 -> ReLU -> Flatten -> fc,
// vae.py:58-61, 72-74): x is the convolution's raw output; no activation tensor, no separate ReLU launches.  Wide fan-in only.
int linear_relu_fwd(const float* x, const float* w, const float* bias, float* y, int B, int In, int Out, hipStream_t st) {
  if (In < 128 || (size_t)B * Out > ((size_t)1 << 22)) return set_error("gpode_linear_relu_fwd: built for wide fan-in layers (In >= 128)");
  hipLaunchKernelGGL(k_linear_fwd_fanin, (unsigned)(((size_t)B * Out + 3) / 4), 256, 0, st, x, w, bias, y, B, In, Out, 1);
  return check_launch("linear_relu_fwd");
}
int linear_relu_bwd(const float* x, const float* w, const float* gy, float* gx, float* gw, float* gb, int B, int In, int Out, hipStream_t st) {
  if (In < 128) return set_error("gpode_linear_relu_bwd: built for wide fan-in layers (In >= 128)");
  if (gx) hipLaunchKernelGGL(k_linear_bwd_x, (unsigned)(((size_t)B * In + 255) / 256), 256, 0, st, gy, w, gx, B, In, Out, x);
  if (gw) hipLaunchKernelGGL(k_linear_bwd_w, (unsigned)(((size_t)Out * (In + 1) * 64 + 255) / 256), 256, 0, st, x, gy, gw, gb, B, In, Out, 1);
  return check_launch("linear_relu_bwd");
}

int loglik_fwd(const float* X, const float* z, float* ll, size_t n, size_t nX, hipStream_t st) {
  hipLaunchKernelGGL(k_loglik_fwd, ew_grid(n), 256, 0, st, X, z, ll, n, nX);
  return check_launch("loglik_fwd");
}
int loglik_bwd(const float* X, const float* z, const float* g, float* gz, size_t n, size_t nX, hipStream_t st) {
  hipLaunchKernelGGL(k_loglik_bwd, ew_grid(n), 256, 0, st, X, z, g, gz, n, nX);
  return check_launch("loglik_bwd");
}
int loglik_rowsum_fwd(const float* X, const float* z, float* out, size_t rows, size_t inner, size_t nX, hipStream_t st) {
  hipLaunchKernelGGL(k_loglik_rowsum, (unsigned)rows, 256, 0, st, X, z, out, inner, nX);
  return check_launch("loglik_rowsum");
}
int loglik_rowsum_bwd(const float* X, const float* z, const float* grow, float* gz, size_t rows, size_t inner, size_t nX, hipStream_t st) {
  const size_t n = rows * inner;
  hipLaunchKernelGGL(k_loglik_rowsum_bwd, ew_grid(n), 256, 0, st, X, z, grow, gz, n, inner, nX);
  return check_launch("loglik_rowsum_bwd");
}


// ---------------------------------------------------------------------------------------------
// ELBO glue on (N, q)-sized tensors, one launch each instead of ~10 elementwise launches apiece:
//   reparameterisation  z = mu + exp(logvar / 2) eps                                   (vae.py:75-78)
//   KL(N(mu, sigma) || N(0, 1)) summed over the latent dimension, per sample          (create_model.py:47-49 via
//       torch.distributions: 0.5 (sigma^2 + mu^2 - 1 - log sigma^2), sigma = exp(logvar / 2))
//   loss = -(mean(lhood) n - mean(kl) n - kl_u), with the three logged terms           (create_model.py:61-73)
// ---------------------------------------------------------------------------------------------
// mu / logvar: (N, q) with row stride ld (ld = q: separate tensors; ld = 2q: the two halves of the encoder's fc output);
// gradients go to gmu / glogvar with row stride ldg
__global__ void k_reparam_fwd(const float* __restrict__ mu, const float* __restrict__ logvar, int ld, const float* __restrict__ eps,
                              float* __restrict__ z, int N, int q) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * q) return;
  const int n = e / q, d = e % q;
  z[e] = mu[(size_t)n * ld + d] + expf(0.5f * logvar[(size_t)n * ld + d]) * eps[e];
}
__global__ void k_reparam_bwd(const float* __restrict__ gz, const float* __restrict__ logvar, int ld, const float* __restrict__ eps,
                              float* __restrict__ gmu, float* __restrict__ glogvar, int ldg, int N, int q) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * q) return;
  const int n = e / q, d = e % q;
  const float g = gz[e];
  gmu[(size_t)n * ldg + d] = g;
  glogvar[(size_t)n * ldg + d] = g * eps[e] * (0.5f * expf(0.5f * logvar[(size_t)n * ld + d]));
}
// z = mu + exp(logvar / 2) eps AND the workgroup's share of sum_{n,d} KL(N(mu, sigma) || N(0, 1)): klpart[blockIdx.x] (summed by the ELBO
// kernel, so the KL term needs no pass of its own over (mu, logvar)); backward: the two gradients of (mu, logvar) -- through z and through the
// KL sum (gkl: the same value for every term) -- in one write, instead of two tensors that autograd then adds
__global__ __launch_bounds__(256) void k_reparam_kl_fwd(const float* __restrict__ mu, const float* __restrict__ logvar, int ld,
                                                         const float* __restrict__ eps, float* __restrict__ z, float* __restrict__ klpart, int N,
                                                         int q) {
  __shared__ float red[4];
  const int e = blockIdx.x * 256 + threadIdx.x;
  float kl = 0.f;
  if (e < N * q) {
    const int n = e / q, d = e % q;
    const float m = mu[(size_t)n * ld + d], sg = expf(0.5f * logvar[(size_t)n * ld + d]);
    z[e] = m + sg * eps[e];
    const float vr = sg * sg;
    kl = 0.5f * (vr + m * m - 1.f - logf(vr));
  }
  const float in1[1] = {kl};
  float out1[1];
  wave_sum_multi<1>(in1, out1);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = out1[0];
  __syncthreads();
  if (threadIdx.x == 0) klpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void k_reparam_kl_bwd(const float* __restrict__ gz, const float* __restrict__ gklpart, const float* __restrict__ mu,
                                 const float* __restrict__ logvar, int ld, const float* __restrict__ eps, float* __restrict__ gmu,
                                 float* __restrict__ glogvar, int ldg, int N, int q) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * q) return;
  const int n = e / q, d = e % q;
  const float g = gz ? gz[e] : 0.f, gk = gklpart ? gklpart[blockIdx.x] : 0.f;
  const float m = mu[(size_t)n * ld + d], lv = logvar[(size_t)n * ld + d];
  gmu[(size_t)n * ldg + d] = g + gk * m;
  glogvar[(size_t)n * ldg + d] = g * eps[e] * (0.5f * expf(0.5f * lv)) + gk * 0.5f * (expf(lv) - 1.f);
}
// one thread per sample
__global__ void k_normal_kl_fwd(const float* __restrict__ mu, const float* __restrict__ logvar, int ld, float* __restrict__ klrow, int N,
                                int q) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float acc = 0.f;
  for (int d = 0; d < q; ++d) {
    const float m = mu[(size_t)n * ld + d], sg = expf(0.5f * logvar[(size_t)n * ld + d]);
    const float vr = sg * sg;                        // var_ratio = (sigma_q / sigma_p)^2, t1 = mu^2 (torch/distributions/kl.py _kl_normal_normal)
    acc += 0.5f * (vr + m * m - 1.f - logf(vr));
  }
  klrow[n] = acc;
}
__global__ void k_normal_kl_bwd(const float* __restrict__ grow, const float* __restrict__ mu, const float* __restrict__ logvar, int ld,
                                float* __restrict__ gmu, float* __restrict__ glogvar, int ldg, int N, int q) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * q) return;
  const int n = e / q, d = e % q;
  const float g = grow[n];
  gmu[(size_t)n * ldg + d] = g * mu[(size_t)n * ld + d];
  glogvar[(size_t)n * ldg + d] = g * 0.5f * (expf(logvar[(size_t)n * ld + d]) - 1.f);
}
// out[0..3] = {loss, -mean lhood, mean kl, kl_u}; one workgroup
__global__ __launch_bounds__(256) void k_elbo_fwd(const float* __restrict__ lhood, int nl, const float* __restrict__ klrow, int nk,
                                                   const float* __restrict__ kl_u, float nobs, float* __restrict__ out) {
  __shared__ float red[4][2];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < nl; i += 256) a += lhood[i];
  for (int i = threadIdx.x; i < nk; i += 256) b += klrow[i];
  const float in2[2] = {a, b};
  float out2[2];
  wave_sum_multi<2>(in2, out2);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = out2[0]; red[threadIdx.x >> 6][1] = out2[1]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float lh = ((red[0][0] + red[1][0]) + (red[2][0] + red[3][0])) / (float)nl;
    const float kr = ((red[0][1] + red[1][1]) + (red[2][1] + red[3][1])) / (float)nk;
    const float ku = kl_u[0];
    out[0] = -(lh * nobs - kr * nobs - ku);
    out[1] = -lh;
    out[2] = kr;
    out[3] = ku;
  }
}
// gout[0..3]: gradients w.r.t. the four outputs -> gradients of the likelihood rows, the KL rows and kl_u
__global__ void k_elbo_bwd(const float* __restrict__ gout, int nl, int nk, float nobs, float* __restrict__ glhood,
                           float* __restrict__ gklrow, float* __restrict__ gklu) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const float g0 = gout[0];
  const float gl = (-g0 * nobs - gout[1]) / (float)nl;
  const float gk = (g0 * nobs + gout[2]) / (float)nk;
  if (e < nl) glhood[e] = gl;
  if (e < nk) gklrow[e] = gk;
  if (e == 0) gklu[0] = g0 + gout[3];
}

// ---- the decoder's output activation fused with the likelihood, and the ELBO assembled in one launch -------------------
// z = sigmoid(a) (vae.py:84, nn.Sigmoid) and sum over a row slice of log(z) X + log(1 - z)(1 - X) (vae.py:136-153 summed as
// create_model.py:49 does): grid (nsplit, rows); part[row][split].  Same expressions as k_act_fwd / k_loglik_rowsum, so z and
// every term are bit-identical to the separate kernels; only the order of the row sum differs.
__global__ __launch_bounds__(256) void k_sigmoid_loglik_fwd(const float* __restrict__ X, const float* __restrict__ a, float* __restrict__ z,
                                                             float* __restrict__ part, size_t inner, size_t nX, size_t chunk) {
  __shared__ float red[4];
  const size_t row = blockIdx.y, p0 = (size_t)blockIdx.x * chunk, p1 = p0 + chunk < inner ? p0 + chunk : inner;
  float acc = 0.f;
  auto one = [&](float av, float xv) {
    const float zv = 1.f / (1.f + expf(-av));
    acc += logf(zv) * xv + logf(1.f - zv) * (1.f - xv);
    return zv;
  };
  if (((inner | nX | chunk) & 3) == 0) {             // rows, wraps of X and slices all start on 16-byte boundaries
    for (size_t p = p0 + 4 * threadIdx.x; p < p1; p += 1024) {
      const size_t e = row * inner + p;
      const float4 av = *reinterpret_cast<const float4*>(a + e), xv = *reinterpret_cast<const float4*>(X + e % nX);
      float4 zv;
      zv.x = one(av.x, xv.x); zv.y = one(av.y, xv.y); zv.z = one(av.z, xv.z); zv.w = one(av.w, xv.w);
      *reinterpret_cast<float4*>(z + e) = zv;
    }
  } else {
    for (size_t p = p0 + threadIdx.x; p < p1; p += 256) {
      const size_t e = row * inner + p;
      z[e] = one(a[e], X[e % nX]);
    }
  }
  float in1[1] = {acc}, out1[1];
  wave_sum_multi<1>(in1, out1);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = out1[0];
  __syncthreads();
  if (threadIdx.x == 0) part[row * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// ga = d/da: k_loglik_rowsum_bwd followed by k_act_bwd (sigmoid), same expressions in the same order
__global__ void k_sigmoid_loglik_bwd(const float* __restrict__ X, const float* __restrict__ z, const float* __restrict__ grow,
                                     float* __restrict__ ga, size_t n, size_t inner, size_t nX) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
    const float xv = X[e % nX], zv = z[e];
    const float gz = grow[e / inner] * (xv / zv - (1.f - xv) / (1.f - zv));
    ga[e] = gz * zv * (1.f - zv);
  }
}

// out[0..3] = {loss, -mean lhood, mean kl, kl_u} (create_model.py:61-73) from the likelihood partial sums (nl_values of them over
// nl_rows rows), the encoder's packed (mu | logvar) rows hs (and hv for second-order models; KL of a factorised Gaussian is additive)
// and the inducing posterior (Um, packed Us: svpy.py:144-175, k_svgp_kl in gp_misc.hip).  ONE workgroup, fixed-order reductions.
// sum of squares of a long vector in kElboParts fixed slices (the packed Us of a big inducing set: 2.1 M entries at M = 512, D = 16
// took one workgroup 0.87 ms)
constexpr int kElboParts = 256;
__global__ __launch_bounds__(256) void k_sumsq_parts(const float* __restrict__ v, size_t n, float* __restrict__ part) {
  __shared__ float red[4];
  const size_t chunk = (n + kElboParts - 1) / kElboParts, e0 = blockIdx.x * chunk, e1 = e0 + chunk < n ? e0 + chunk : n;
  float acc = 0.f;
  for (size_t e = e0 + threadIdx.x; e < e1; e += 256) acc = fmaf(v[e], v[e], acc);
  float in1[1] = {acc}, out1[1];
  wave_sum_multi<1>(in1, out1);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = out1[0];
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(1024) void k_elbo_all_fwd(const float* __restrict__ lpart, int nl_rows, int nl_values, const float* __restrict__ hs,
                                                        const float* __restrict__ hv, int N, int q, int M, int Do,
                                                        const float* __restrict__ Um, const float* __restrict__ Us, float nobs,
                                                        float* __restrict__ out, const float* __restrict__ usq_part,
                                                        const float* __restrict__ kls = nullptr, int nks = 0,
                                                        const float* __restrict__ klv = nullptr, int nkv = 0) {
  __shared__ float red[16][3];
  const size_t P = (size_t)M * (M + 1) / 2;
  float u = 0.f, a = 0.f, b = 0.f;
  // one workgroup walks 30 000 + 16 000 values: eight independent loads per thread in flight instead of one per dependent add
  auto strided = [&](const float* __restrict__ p, size_t n, bool square) {
    const size_t bd = blockDim.x;
    float sacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    size_t e = threadIdx.x;
    for (; e + 7 * bd < n; e += 8 * bd) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = p[e + k * bd];
#pragma unroll
      for (int k = 0; k < 8; ++k) sacc[k] = square ? fmaf(v[k], v[k], sacc[k]) : sacc[k] + v[k];
    }
    for (; e < n; e += bd) sacc[0] = square ? fmaf(p[e], p[e], sacc[0]) : sacc[0] + p[e];
    return ((sacc[0] + sacc[1]) + (sacc[2] + sacc[3])) + ((sacc[4] + sacc[5]) + (sacc[6] + sacc[7]));
  };
  if (usq_part) {                                    // ||Us||_F^2 arrives as kElboParts partial sums
    for (int e = threadIdx.x; e < kElboParts; e += blockDim.x) u += usq_part[e];
  } else {
    u = strided(Us, P * Do, true);
  }
  for (int e = threadIdx.x; e < M * Do; e += blockDim.x) {
    const int m = e / Do, d = e % Do;
    const float l = Us[(size_t)d * P + (size_t)m * (m + 1) / 2 + m];
    const float v = Um[e];
    u += v * v - logf(l * l);
  }
  a = strided(lpart, (size_t)nl_values, false);
  // hs == nullptr: the KL terms arrive summed per workgroup of k_reparam_kl_fwd (kls / klv)
  if (nks > 0) b += strided(kls, (size_t)nks, false);
  if (nkv > 0) b += strided(klv, (size_t)nkv, false);
  for (int e = threadIdx.x; hs && e < N * q; e += blockDim.x) {
    const int n = e / q, d = e % q;
    {
      const float m = hs[(size_t)n * 2 * q + d], sg = expf(0.5f * hs[(size_t)n * 2 * q + q + d]);
      const float vr = sg * sg;
      b += 0.5f * (vr + m * m - 1.f - logf(vr));
    }
    if (hv) {
      const float m = hv[(size_t)n * 2 * q + d], sg = expf(0.5f * hv[(size_t)n * 2 * q + q + d]);
      const float vr = sg * sg;
      b += 0.5f * (vr + m * m - 1.f - logf(vr));
    }
  }
  const float in3[3] = {u, a, b};
  float out3[3];
  wave_sum_multi<3>(in3, out3);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = out3[0]; red[threadIdx.x >> 6][1] = out3[1]; red[threadIdx.x >> 6][2] = out3[2]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[3] = {0.f, 0.f, 0.f};
    for (int w = 0; w < 16; ++w) { t[0] += red[w][0]; t[1] += red[w][1]; t[2] += red[w][2]; }
    const float ku = 0.5f * (t[0] - (float)M * (float)Do);
    const float lh = t[1] / (float)nl_rows, kr = t[2] / (float)N;
    out[0] = -(lh * nobs - kr * nobs - ku);
    out[1] = -lh;
    out[2] = kr;
    out[3] = ku;
  }
}
// g0..g3: gradients w.r.t. the four outputs (device scalars, NULL = 0) -> likelihood rows, packed encoder rows, Um, Us
__global__ void k_elbo_all_bwd(const float* __restrict__ g0p, const float* __restrict__ g1p, const float* __restrict__ g2p,
                               const float* __restrict__ g3p, int nl_rows, const float* __restrict__ hs, const float* __restrict__ hv, int N,
                               int q, int M, int Do, const float* __restrict__ Um, const float* __restrict__ Us, float nobs,
                               float* __restrict__ glrow, float* __restrict__ ghs, float* __restrict__ ghv, float* __restrict__ dUm,
                               float* __restrict__ dUs) {
  const size_t P = (size_t)M * (M + 1) / 2;
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const float g0 = g0p ? *g0p : 0.f, g1 = g1p ? *g1p : 0.f, g2 = g2p ? *g2p : 0.f, g3 = g3p ? *g3p : 0.f;
  if (e < (size_t)nl_rows) glrow[e] = (-g0 * nobs - g1) / (float)nl_rows;
  if (e < (size_t)N * q) {
    const float gk = (g0 * nobs + g2) / (float)N;
    const int n = (int)(e / q), d = (int)(e % q);
    const size_t im = (size_t)n * 2 * q + d, il = im + q;
    ghs[im] = gk * hs[im];
    ghs[il] = gk * 0.5f * (expf(hs[il]) - 1.f);
    if (hv) {
      ghv[im] = gk * hv[im];
      ghv[il] = gk * 0.5f * (expf(hv[il]) - 1.f);
    }
  }
  const float g = g0 + g3;
  if (e < (size_t)M * Do) dUm[e] = g * Um[e];
  if (e < P * Do) {
    const size_t k = e % P;
    int n = (int)((sqrtf(8.f * (float)k + 1.f) - 1.f) * 0.5f);
    while ((size_t)(n + 1) * (n + 2) / 2 <= k) ++n;
    while ((size_t)n * (n + 1) / 2 > k) --n;
    const bool diag = (k - (size_t)n * (n + 1) / 2) == (size_t)n;
    const float v = Us[e];
    dUs[e] = g * (diag ? v - 1.f / v : v);
  }
}

// k_elbo_all_bwd and k_sigmoid_loglik_bwd in ONE launch: every likelihood row receives the same gradient (-g0 nobs - g1) / rows, so the
// logit gradients need no row-gradient tensor in between -- blocks [0, nb_small) do the (N,q)- and (M,D)-sized outputs, the rest the logits.
__global__ void k_elbo_loglik_bwd(const float* __restrict__ g0p, const float* __restrict__ g1p, const float* __restrict__ g2p,
                                  const float* __restrict__ g3p, int nl_rows, const float* __restrict__ hs, const float* __restrict__ hv, int N,
                                  int q, int M, int Do, const float* __restrict__ Um, const float* __restrict__ Us, float nobs,
                                  float* __restrict__ glrow, float* __restrict__ ghs, float* __restrict__ ghv, float* __restrict__ dUm,
                                  float* __restrict__ dUs, int nb_small, const float* __restrict__ X, const float* __restrict__ z,
                                  float* __restrict__ ga, size_t n, size_t nX, float* __restrict__ gkls = nullptr, int nks = 0,
                                  float* __restrict__ gklv = nullptr, int nkv = 0) {
  const float g0 = g0p ? *g0p : 0.f, g1 = g1p ? *g1p : 0.f, g2 = g2p ? *g2p : 0.f, g3 = g3p ? *g3p : 0.f;
  if ((int)blockIdx.x < nb_small) {
    const size_t P = (size_t)M * (M + 1) / 2;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < (size_t)nl_rows) glrow[e] = (-g0 * nobs - g1) / (float)nl_rows;
    if (e < (size_t)nks) gkls[e] = (g0 * nobs + g2) / (float)N;         // hs == nullptr: d loss / d (KL partial sum), the same for all
    if (e < (size_t)nkv) gklv[e] = (g0 * nobs + g2) / (float)N;
    if (hs && e < (size_t)N * q) {
      const float gk = (g0 * nobs + g2) / (float)N;
      const int nn = (int)(e / q), d = (int)(e % q);
      const size_t im = (size_t)nn * 2 * q + d, il = im + q;
      ghs[im] = gk * hs[im];
      ghs[il] = gk * 0.5f * (expf(hs[il]) - 1.f);
      if (hv) {
        ghv[im] = gk * hv[im];
        ghv[il] = gk * 0.5f * (expf(hv[il]) - 1.f);
      }
    }
    const float g = g0 + g3;
    if (e < (size_t)M * Do) dUm[e] = g * Um[e];
    if (e < P * Do) {
      const size_t k = e % P;
      int r = (int)((sqrtf(8.f * (float)k + 1.f) - 1.f) * 0.5f);
      while ((size_t)(r + 1) * (r + 2) / 2 <= k) ++r;
      while ((size_t)r * (r + 1) / 2 > k) --r;
      const bool diag = (k - (size_t)r * (r + 1) / 2) == (size_t)r;
      const float v = Us[e];
      dUs[e] = g * (diag ? v - 1.f / v : v);
    }
    return;
  }
  const float gl = (-g0 * nobs - g1) / (float)nl_rows;
  const size_t nb = gridDim.x - nb_small;
  for (size_t e = ((size_t)blockIdx.x - nb_small) * blockDim.x + threadIdx.x; e < n; e += nb * blockDim.x) {
    const float xv = X[e % nX], zv = z[e];
    const float gz = gl * (xv / zv - (1.f - xv) / (1.f - zv));
    ga[e] = gz * zv * (1.f - zv);
  }
}

int sigmoid_loglik_splits(size_t rows, size_t inner) {
  size_t ns = (1024 + rows - 1) / rows, cap = (inner + 1023) / 1024;
  if (ns > cap) ns = cap;
  if (ns > 64) ns = 64;
  return ns < 1 ? 1 : (int)ns;
}
int sigmoid_loglik_fwd(const float* X, const float* a, float* z, float* part, size_t rows, size_t inner, size_t nX, int nsplit,
                       hipStream_t st) {
  if (nsplit < 1 || rows > 65535) return set_error("gpode_sigmoid_loglik_fwd: nsplit >= 1, rows <= 65535");
  if ((rows * inner) % nX != 0) return set_error("gpode_sigmoid_loglik_fwd: X must tile the rows");
  size_t chunk = (inner + nsplit - 1) / nsplit;
  chunk = (chunk + 3) & ~(size_t)3;
  if (((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(z)) & 15) != 0) chunk |= 1;   // scalar path
  hipLaunchKernelGGL(k_sigmoid_loglik_fwd, dim3(nsplit, (unsigned)rows), 256, 0, st, X, a, z, part, inner, nX, chunk);
  return check_launch("sigmoid_loglik_fwd");
}
int sigmoid_loglik_bwd(const float* X, const float* z, const float* grow, float* ga, size_t rows, size_t inner, size_t nX, hipStream_t st) {
  const size_t n = rows * inner;
  hipLaunchKernelGGL(k_sigmoid_loglik_bwd, ew_grid(n), 256, 0, st, X, z, grow, ga, n, inner, nX);
  return check_launch("sigmoid_loglik_bwd");
}
int elbo_all_fwd(const float* lpart, int nl_rows, int nl_values, const float* hs, const float* hv, int N, int q, int M, int Do,
                 const float* Um, const float* Us, float nobs, float* out, hipStream_t st) {
  // out: 4 results + kElboParts floats of scratch
  const size_t nus = (size_t)M * (M + 1) / 2 * Do;
  const float* usq = nullptr;
  if (nus > ((size_t)1 << 16)) {
    hipLaunchKernelGGL(k_sumsq_parts, kElboParts, 256, 0, st, Us, nus, out + 4);
    usq = out + 4;
  }
  hipLaunchKernelGGL(k_elbo_all_fwd, 1, 1024, 0, st, lpart, nl_rows, nl_values, hs, hv, N, q, M, Do, Um, Us, nobs, out, usq, (const float*)nullptr, 0, (const float*)nullptr, 0);
  return check_launch("elbo_all_fwd");
}
int elbo_all_bwd(const float* g0, const float* g1, const float* g2, const float* g3, int nl_rows, const float* hs, const float* hv, int N,
                 int q, int M, int Do, const float* Um, const float* Us, float nobs, float* glrow, float* ghs, float* ghv, float* dUm,
                 float* dUs, hipStream_t st) {
  size_t n = (size_t)M * (M + 1) / 2 * Do;
  if ((size_t)N * q > n) n = (size_t)N * q;
  if ((size_t)nl_rows > n) n = nl_rows;
  if ((size_t)M * Do > n) n = (size_t)M * Do;
  hipLaunchKernelGGL(k_elbo_all_bwd, (unsigned)((n + 255) / 256), 256, 0, st, g0, g1, g2, g3, nl_rows, hs, hv, N, q, M, Do, Um, Us, nobs, glrow,
                     ghs, ghv, dUm, dUs);
  return check_launch("elbo_all_bwd");
}

int elbo_all_bwd_ll(const float* g0, const float* g1, const float* g2, const float* g3, int nl_rows, const float* hs, const float* hv, int N,
                    int q, int M, int Do, const float* Um, const float* Us, float nobs, float* glrow, float* ghs, float* ghv, float* dUm,
                    float* dUs, const float* X, const float* z, float* ga, size_t n, size_t nX, hipStream_t st) {
  size_t ns = (size_t)M * (M + 1) / 2 * Do;
  if ((size_t)N * q > ns) ns = (size_t)N * q;
  if ((size_t)nl_rows > ns) ns = nl_rows;
  if ((size_t)M * Do > ns) ns = (size_t)M * Do;
  const int nb_small = (int)((ns + 255) / 256);
  hipLaunchKernelGGL(k_elbo_loglik_bwd, (unsigned)(nb_small + ew_grid(n)), 256, 0, st, g0, g1, g2, g3, nl_rows, hs, hv, N, q, M, Do, Um, Us, nobs,
                     glrow, ghs, ghv, dUm, dUs, nb_small, X, z, ga, n, nX, (float*)nullptr, 0, (float*)nullptr, 0);
  return check_launch("elbo_loglik_bwd");
}

int reparam_kl_fwd(const float* mu, const float* logvar, int ld, const float* eps, float* z, float* klpart, int N, int q, hipStream_t st) {
  hipLaunchKernelGGL(k_reparam_kl_fwd, (N * q + 255) / 256, 256, 0, st, mu, logvar, ld, eps, z, klpart, N, q);
  return check_launch("reparam_kl_fwd");
}
int reparam_kl_bwd(const float* gz, const float* gklpart, const float* mu, const float* logvar, int ld, const float* eps, float* gmu,
                   float* glogvar, int ldg, int N, int q, hipStream_t st) {
  hipLaunchKernelGGL(k_reparam_kl_bwd, (N * q + 255) / 256, 256, 0, st, gz, gklpart, mu, logvar, ld, eps, gmu, glogvar, ldg, N, q);
  return check_launch("reparam_kl_bwd");
}
int elbo_all_fwd_kl(const float* lpart, int nl_rows, int nl_values, const float* kls, int nks, const float* klv, int nkv, int N, int M, int Do,
                    const float* Um, const float* Us, float nobs, float* out, hipStream_t st) {
  const size_t nus = (size_t)M * (M + 1) / 2 * Do;
  const float* usq = nullptr;
  if (nus > ((size_t)1 << 16)) {
    hipLaunchKernelGGL(k_sumsq_parts, kElboParts, 256, 0, st, Us, nus, out + 4);
    usq = out + 4;
  }
  hipLaunchKernelGGL(k_elbo_all_fwd, 1, 1024, 0, st, lpart, nl_rows, nl_values, (const float*)nullptr, (const float*)nullptr, N, 1, M, Do, Um, Us, nobs,
                     out, usq, kls, nks, klv, nkv);
  return check_launch("elbo_all_fwd_kl");
}
int elbo_all_bwd_ll_kl(const float* g0, const float* g1, const float* g2, const float* g3, int nl_rows, int N, int M, int Do, const float* Um,
                       const float* Us, float nobs, float* glrow, float* gkls, int nks, float* gklv, int nkv, float* dUm, float* dUs,
                       const float* X, const float* z, float* ga, size_t n, size_t nX, hipStream_t st) {
  size_t ns = (size_t)M * (M + 1) / 2 * Do;
  if ((size_t)nl_rows > ns) ns = nl_rows;
  if ((size_t)M * Do > ns) ns = (size_t)M * Do;
  if ((size_t)nks > ns) ns = nks;
  if ((size_t)nkv > ns) ns = nkv;
  const int nb_small = (int)((ns + 255) / 256);
  float* nf = nullptr;
  hipLaunchKernelGGL(k_elbo_loglik_bwd, (unsigned)(nb_small + ew_grid(n)), 256, 0, st, g0, g1, g2, g3, nl_rows, (const float*)nullptr,
                     (const float*)nullptr, N, 1, M, Do, Um, Us, nobs, glrow, nf, nf, dUm, dUs, nb_small, X, z, ga, n, nX, gkls, nks, gklv, nkv);
  return check_launch("elbo_loglik_bwd_kl");
}
int reparam_fwd(const float* mu, const float* logvar, int ld, const float* eps, float* z, int N, int q, hipStream_t st) {
  hipLaunchKernelGGL(k_reparam_fwd, (N * q + 255) / 256, 256, 0, st, mu, logvar, ld, eps, z, N, q);
  return check_launch("reparam_fwd");
}
int reparam_bwd(const float* gz, const float* logvar, int ld, const float* eps, float* gmu, float* glogvar, int ldg, int N, int q,
                hipStream_t st) {
  hipLaunchKernelGGL(k_reparam_bwd, (N * q + 255) / 256, 256, 0, st, gz, logvar, ld, eps, gmu, glogvar, ldg, N, q);
  return check_launch("reparam_bwd");
}
int normal_kl_fwd(const float* mu, const float* logvar, int ld, float* klrow, int N, int q, hipStream_t st) {
  hipLaunchKernelGGL(k_normal_kl_fwd, (N + 255) / 256, 256, 0, st, mu, logvar, ld, klrow, N, q);
  return check_launch("normal_kl_fwd");
}
int normal_kl_bwd(const float* grow, const float* mu, const float* logvar, int ld, float* gmu, float* glogvar, int ldg, int N, int q,
                  hipStream_t st) {
  hipLaunchKernelGGL(k_normal_kl_bwd, (N * q + 255) / 256, 256, 0, st, grow, mu, logvar, ld, gmu, glogvar, ldg, N, q);
  return check_launch("normal_kl_bwd");
}
int elbo_fwd(const float* lhood, int nl, const float* klrow, int nk, const float* kl_u, float nobs, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_elbo_fwd, 1, 256, 0, st, lhood, nl, klrow, nk, kl_u, nobs, out);
  return check_launch("elbo_fwd");
}
int elbo_bwd(const float* gout, int nl, int nk, float nobs, float* glhood, float* gklrow, float* gklu, hipStream_t st) {
  const int n = nl > nk ? nl : nk;
  hipLaunchKernelGGL(k_elbo_bwd, (n + 255) / 256, 256, 0, st, gout, nl, nk, nobs, glhood, gklrow, gklu);
  return check_launch("elbo_bwd");
}

}  // namespace gp

// ---------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults as used at main.py:194: betas (0.9, 0.999), eps 1e-8, no weight decay),
// all parameter tensors in ONE launch: a table of (param, grad, exp_avg, exp_avg_sq) pointers with prefix
// offsets; each thread finds its tensor by binary search.
// ---------------------------------------------------------------------------------------------
namespace gp {
__global__ void k_adam_multi(float* const* __restrict__ params, const float* const* __restrict__ grads,
                             float* const* __restrict__ m1, float* const* __restrict__ m2, const long long* __restrict__ offs,
                             int ntensors, long long total, float lr, float beta1, float beta2, float eps, float bc1, float bc2,
                             int* __restrict__ step_dev) {
  int step = 0;
  if (step_dev) {                                    // step counter kept on the device (captured graphs replay with a live count)
    step = step_dev[0] + 1;                          // this update's number; published below by the last workgroup to finish
    const float t = (float)step;
    bc1 = 1.f - powf(beta1, t);
    bc2 = 1.f - powf(beta2, t);
  }
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = ntensors - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (offs[mid] <= e) lo = mid; else hi = mid - 1; }
    const long long i = e - offs[lo];
    const float g = grads[lo][i];
    const float a = beta1 * m1[lo][i] + (1.f - beta1) * g;
    const float b = beta2 * m2[lo][i] + (1.f - beta2) * g * g;
    m1[lo][i] = a;
    m2[lo][i] = b;
    // torch: denom = sqrt(v)/sqrt(bc2) + eps ; p -= lr/bc1 * m / denom
    params[lo][i] -= (lr / bc1) * a / (sqrtf(b) / sqrtf(bc2) + eps);
  }
  if (step_dev) {
    // every workgroup has read step_dev[0] before it takes a ticket, so the last one may advance it (no separate launch for the
    // counter: one graph node less on the tail of every step)
    // (no memory fence: nothing but the counter is handed between workgroups, and the ticket is a device-scope atomic; a
    // __threadfence() here writes the L2 back -- measured 15 us on this 22 us kernel)
    __syncthreads();
    if (threadIdx.x == 0) {
      if (atomicAdd(&step_dev[1], 1) == (int)gridDim.x - 1) {
        step_dev[1] = 0;
        step_dev[0] = step;
      }
    }
  }
}

// flat[offs[t] + i] = grads[t][i] for every tensor of the table: the data-parallel gradient bucket filled in ONE launch
// from the tensors autograd handed over (instead of one accumulate kernel per parameter into persistent views)
__global__ void k_gather_multi(const float* const* __restrict__ grads, const long long* __restrict__ offs, int ntensors,
                               long long total, float* __restrict__ flat) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = ntensors - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (offs[mid] <= e) lo = mid; else hi = mid - 1; }
    flat[e] = grads[lo][e - offs[lo]];
  }
}

int gather_multi(const float* const* grads, const long long* offs, int ntensors, long long total, float* flat, hipStream_t st) {
  hipLaunchKernelGGL(k_gather_multi, ew_grid((size_t)total), 256, 0, st, grads, offs, ntensors, total, flat);
  return check_launch("gather_multi");
}

// step_dev != nullptr: int[2] {step count, 0}: the count is advanced by the kernel and used as this update's number (host `step` ignored)
int adam_multi(float* const* params, const float* const* grads, float* const* m1, float* const* m2, const long long* offs,
               int ntensors, long long total, float lr, float beta1, float beta2, float eps, int step, int* step_dev, hipStream_t st) {
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  // with the device-side count every workgroup takes a ticket on ONE address (the last one publishes the count).  The cap used to be
  // 128 workgroups "to keep the tickets few"; an iteration of the grid-stride loop is a chain of dependent loads (binary search of the
  // offsets, the tensor's pointers, its data: ~1.5 us), and the divergence-free model's 0.6 M parameters made that 18 iterations per
  // thread.  2048 workgroups: configs[1] 2.62 -> 2.59 ms on one box and no change on another, configs[2] / [3] -8 us, configs[0]
  // unchanged (GPODE_ADAM_MAX_WG, A/B in profiles/r03d_ab_switches.txt)
  unsigned grid = ew_grid((size_t)total);
  static const unsigned cap = [] { const char* e = getenv("GPODE_ADAM_MAX_WG"); return e ? (unsigned)atoi(e) : 2048u; }();
  if (step_dev && grid > cap) grid = cap;
  hipLaunchKernelGGL(k_adam_multi, grid, 256, 0, st, params, grads, m1, m2, offs, ntensors, total, lr, beta1, beta2, eps, bc1, bc2,
                     step_dev);
  return check_launch("adam_multi");
}
}  // namespace gp

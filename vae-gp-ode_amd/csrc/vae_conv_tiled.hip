// vae_conv_tiled.hip -- layer-specialised kernels and dispatch for the VAE's convolutions (vae.py:52-59, 64-84).
//
// The production path is on the fp32 matrix cores: the implicit-GEMM engine of conv_mfma.hpp (decnn.1/4/7 forward, d/d input,
// d/d weight; the encoder's cnn.3 / cnn.6 where the channel counts fill MFMA tiles) and conv_dec10_mfma.hpp (decnn.10 with
// the 25 taps as the GEMM dimension).  This file holds the layer geometry (CTLayer), the dispatch by geometry
// (tiled_fwd / tiled_bwd_data / tiled_bwd_weight; -1 = no specialisation, the caller falls back to the generic direct
// kernels of vae_conv.hip) and the earlier LDS-resident VALU kernels, kept selectable with GPODE_CONV_VALU=1 as the
// A/B baseline the MFMA numbers in DESIGN.md are quoted against:
//
//   a workgroup stages a few whole images (zero-padded, so taps never branch) and a slab of weights in LDS; a thread
//   owns ONE pixel x 16 channels of output in registers.  Per (input channel, tap) it reads one input value
//   (ds_read_b32, lane = pixel -> conflict-free) and 16 weights (4 x ds_read_b128 from a [tap][ci][co] slab: every
//   lane of a channel group reads the same address -> broadcast) and issues 16 FMAs.
//   T1 convT_fwd   gather form, one stride-parity class of output pixels at a time: inside a class every
//                  pixel sees the same taps, so the weight reads are wave-uniform
//   T2 convT_bwd   d/d input = ordinary strided convolution of grad_output with the same weights
//   T3 convT_wgrad d/d weight: thread = (input channel, 4 output channels) x 25 taps = 100 accumulators,
//                  images streamed through LDS; batch split over workgroups, slabs summed in fixed order
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdint>
#include "gp_launch.hpp"
#include <type_traits>
#include "conv_layers.hpp"
#include "conv_mfma.hpp"
#include "conv_dec10_mfma.hpp"
#include "conv_dec1_mfma.hpp"
#include "conv_dec4_mfma.hpp"

namespace gp {



extern __shared__ __attribute__((aligned(16))) float tsm[];

// ---------------------------------------------------------------------------------------------
// T1: y[b,co,oy,ox] = bias[co] + sum_{ci,ky,kx} x[b,ci,iy,ix] w[ci,co,ky,kx],  oy = iy*2 - 1 + ky
//     grid.x = image groups of IPB, block 256.
// ---------------------------------------------------------------------------------------------
template <class L, int IPB>
__global__ __launch_bounds__(256) void k_convT_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ y, int B) {
  constexpr int CI = L::CI, CO = L::CO, HI = L::HI, HO = L::HO, HP = L::HP, K = L::K, S = L::S, P = L::P, PL = L::PL;
  constexpr int CPT = CO % 16 == 0 ? 16 : 8;       // output channels per thread
  static_assert(CO % CPT == 0, "8 or 16 output channels per thread");
  constexpr int NG = CO / CPT;                     // channel groups per pixel
  constexpr int IMG = CI * HP * HP;                // floats per staged image
  float* s_img = tsm;                              // [IPB][CI][HP][HP]
  float* s_w = tsm + IPB * IMG;                    // [tap][CI][CO] for the current parity class (<= 9 taps)
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * IPB;
  const int nimg = min(IPB, B - b0);
  // stage the images, zero padded
  for (int e = tid; e < IPB * IMG; e += 256) {
    const int im = e / IMG, r = e % IMG, ci = r / (HP * HP), yy = (r / HP) % HP - PL, xx = r % HP - PL;
    float v = 0.f;
    if (im < nimg && yy >= 0 && yy < HI && xx >= 0 && xx < HI) v = x[(((size_t)(b0 + im) * CI + ci) * HI + yy) * HI + xx];
    s_img[e] = v;
  }
  for (int cls = 0; cls < S * S; ++cls) {
    const int py = cls / S, px = cls % S;          // (oy + P) % S, (ox + P) % S
    const int nty = (K - py + S - 1) / S, ntx = (K - px + S - 1) / S;   // taps ky = py + S t
    __syncthreads();                               // previous class done with s_w (and images staged)
    for (int e = tid; e < nty * ntx * CI * CO; e += 256) {
      const int co = e % CO, ci = (e / CO) % CI, tap = e / (CO * CI);
      const int ky = py + S * (tap / ntx), kx = px + S * (tap % ntx);
      s_w[e] = w[(((size_t)ci * CO + co) * K + ky) * K + kx];
    }
    __syncthreads();
    // pixels of the class: oy = S qy + py - P in [0, HO)
    const int qy0 = (P - py + S - 1 > 0) ? (P - py + S - 1) / S : 0, qx0 = (P - px + S - 1 > 0) ? (P - px + S - 1) / S : 0;
    const int ny = (HO - 1 + P - py) / S - qy0 + 1, nx = (HO - 1 + P - px) / S - qx0 + 1;
    const int items = nimg * ny * nx * NG;
    for (int it = tid; it < items; it += 256) {
      const int g = it % NG, p = (it / NG) % (ny * nx), im = it / (NG * ny * nx);
      const int qy = qy0 + p / nx, qx = qx0 + p % nx;
      float acc[CPT];
#pragma unroll
      for (int c = 0; c < CPT; ++c) acc[c] = bias ? bias[g * CPT + c] : 0.f;
      const float* img = s_img + im * IMG;
      for (int ty = 0; ty < nty; ++ty) {
        for (int tx = 0; tx < ntx; ++tx) {
          const float* ip = img + (qy - ty + PL) * HP + (qx - tx + PL);    // iy = qy - ty
          const float4* wp = reinterpret_cast<const float4*>(s_w + ((ty * ntx + tx) * CI) * CO + g * CPT);
#pragma unroll 4
          for (int ci = 0; ci < CI; ++ci) {
            const float v = ip[ci * HP * HP];
#pragma unroll
            for (int q = 0; q < CPT / 4; ++q) {
              const float4 wq = wp[ci * (CO / 4) + q];
              acc[4 * q] = fmaf(v, wq.x, acc[4 * q]); acc[4 * q + 1] = fmaf(v, wq.y, acc[4 * q + 1]);
              acc[4 * q + 2] = fmaf(v, wq.z, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(v, wq.w, acc[4 * q + 3]);
            }
          }
        }
      }
      const int oy = S * qy + py - P, ox = S * qx + px - P;
      float* yp = y + (((size_t)(b0 + im) * CO + g * CPT) * HO + oy) * HO + ox;
#pragma unroll
      for (int c = 0; c < CPT; ++c) yp[(size_t)c * HO * HO] = acc[c];
    }
  }
}


// ---------------------------------------------------------------------------------------------
// T2: gx[b,ci,iy,ix] = sum_{co,ky,kx} gy[b,co,2 iy - 1 + ky, 2 ix - 1 + kx] w[ci,co,ky,kx]
//     weights staged in chunks of COC output channels ([co][tap][ci] slab).
// ---------------------------------------------------------------------------------------------
template <class L, int IPB, int COC>
__global__ __launch_bounds__(256) void k_convT_bwd_data(const float* __restrict__ gy, const float* __restrict__ w,
                                                         float* __restrict__ gx, int B) {
  constexpr int CI = L::CI, CO = L::CO, HI = L::HI, HO = L::HO, GPD = L::GP_, K = L::K, S = L::S, P = L::P, KK = K * K;
  static_assert(CI % 16 == 0 && CO % COC == 0, "blocking");
  constexpr int NG = CI / 16;
  constexpr int IMG = CO * GPD * GPD;
  float* s_img = tsm;                              // [IPB][CO][GPD][GPD], index = oy + P
  float* s_w = tsm + IPB * IMG;                    // [COC][K*K][CI]
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * IPB;
  const int nimg = min(IPB, B - b0);
  for (int e = tid; e < IPB * IMG; e += 256) {
    const int im = e / IMG, r = e % IMG, co = r / (GPD * GPD), yy = (r / GPD) % GPD - P, xx = r % GPD - P;
    float v = 0.f;
    if (im < nimg && yy >= 0 && yy < HO && xx >= 0 && xx < HO) v = gy[(((size_t)(b0 + im) * CO + co) * HO + yy) * HO + xx];
    s_img[e] = v;
  }
  const int items = nimg * HI * HI * NG;
  constexpr int MAXIT = (IPB * HI * HI * NG + 255) / 256;
  float acc[MAXIT][16];
#pragma unroll
  for (int r = 0; r < MAXIT; ++r)
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[r][c] = 0.f;
  for (int c0 = 0; c0 < CO; c0 += COC) {
    __syncthreads();
    for (int e = tid; e < COC * KK * CI; e += 256) {
      const int ci = e % CI, tap = (e / CI) % KK, co = c0 + e / (CI * KK);
      s_w[e] = w[((size_t)ci * CO + co) * KK + tap];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MAXIT; ++r) {
      const int it = tid + 256 * r;
      if (it < items) {
        const int g = it % NG, p = (it / NG) % (HI * HI), im = it / (NG * HI * HI);
        const int iy = p / HI, ix = p % HI;
        const float* img = s_img + im * IMG + (c0 * GPD + S * iy) * GPD + S * ix;   // (oy + P) = S iy + ky
        for (int co = 0; co < COC; ++co) {
#pragma unroll
          for (int ky = 0; ky < K; ++ky) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
              const float v = img[(co * GPD + ky) * GPD + kx];
              const float4* wp = reinterpret_cast<const float4*>(s_w + ((co * KK + ky * K + kx) * CI) + g * 16);
              const float4 w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
              float* a = acc[r];
              a[0] = fmaf(v, w0.x, a[0]); a[1] = fmaf(v, w0.y, a[1]); a[2] = fmaf(v, w0.z, a[2]); a[3] = fmaf(v, w0.w, a[3]);
              a[4] = fmaf(v, w1.x, a[4]); a[5] = fmaf(v, w1.y, a[5]); a[6] = fmaf(v, w1.z, a[6]); a[7] = fmaf(v, w1.w, a[7]);
              a[8] = fmaf(v, w2.x, a[8]); a[9] = fmaf(v, w2.y, a[9]); a[10] = fmaf(v, w2.z, a[10]); a[11] = fmaf(v, w2.w, a[11]);
              a[12] = fmaf(v, w3.x, a[12]); a[13] = fmaf(v, w3.y, a[13]); a[14] = fmaf(v, w3.z, a[14]); a[15] = fmaf(v, w3.w, a[15]);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < MAXIT; ++r) {
    const int it = tid + 256 * r;
    if (it < items) {
      const int g = it % NG, p = (it / NG) % (HI * HI), im = it / (NG * HI * HI);
      float* gp = gx + (((size_t)(b0 + im) * CI + g * 16) * HI * HI) + p;
#pragma unroll
      for (int c = 0; c < 16; ++c) gp[(size_t)c * HI * HI] = acc[r][c];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// T3: gw[ci,co,ky,kx] = sum_{b,iy,ix} x[b,ci,iy,ix] gy[b,co,2 iy - 1 + ky, 2 ix - 1 + kx]
//     grid (nsplit, CO/COW): a workgroup owns COW output channels and a slice of the batch; thread = (ci, 4 co).
//     part[split][ci][co][25]
// ---------------------------------------------------------------------------------------------
template <class L, int COW>
__global__ __launch_bounds__(256) void k_convT_wgrad(const float* __restrict__ x, const float* __restrict__ gy,
                                                      float* __restrict__ part, int B, int b_per_split) {
  constexpr int CI = L::CI, CO = L::CO, HI = L::HI, HO = L::HO, GPD = L::GP_, K = L::K, S = L::S, P = L::P, KK = K * K;
  constexpr int NQ = COW / 4;                      // co-quads per workgroup
  constexpr int NT = CI * NQ;                      // active (ci, quad) threads
  constexpr int PSPLIT = 256 / NT;                 // pixel-parity split when there are spare threads
  static_assert(NT <= 256 && 256 % NT == 0, "thread mapping");
  float* s_x = tsm;                                // [HI*HI][CI]
  float* s_g = tsm + HI * HI * CI;                 // [GPD][GPD][COW]
  const int tid = threadIdx.x;
  const int ci = tid % CI, q = (tid / CI) % NQ, ps = tid / NT;
  const int co0 = blockIdx.y * COW;
  const int b0 = blockIdx.x * b_per_split, b1 = min(B, b0 + b_per_split);
  float acc[KK][4];
#pragma unroll
  for (int t = 0; t < KK; ++t) { acc[t][0] = 0.f; acc[t][1] = 0.f; acc[t][2] = 0.f; acc[t][3] = 0.f; }
  for (int b = b0; b < b1; ++b) {
    __syncthreads();
    for (int e = tid; e < HI * HI * CI; e += 256) {
      const int c = e % CI, p = e / CI;
      s_x[e] = x[((size_t)b * CI + c) * HI * HI + p];
    }
    for (int e = tid; e < GPD * GPD * COW; e += 256) {
      const int c = e % COW, xx = (e / COW) % GPD - P, yy = e / (COW * GPD) - P;
      float v = 0.f;
      if (yy >= 0 && yy < HO && xx >= 0 && xx < HO) v = gy[(((size_t)b * CO + co0 + c) * HO + yy) * HO + xx];
      s_g[e] = v;
    }
    __syncthreads();
    for (int p = ps; p < HI * HI; p += PSPLIT) {
      const int iy = p / HI, ix = p % HI;
      const float xv = s_x[p * CI + ci];
      const float* gp = s_g + ((S * iy) * GPD + S * ix) * COW + q * 4;
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float4 g4 = *reinterpret_cast<const float4*>(gp + (ky * GPD + kx) * COW);
          float* a = acc[ky * K + kx];
          a[0] = fmaf(xv, g4.x, a[0]); a[1] = fmaf(xv, g4.y, a[1]); a[2] = fmaf(xv, g4.z, a[2]); a[3] = fmaf(xv, g4.w, a[3]);
        }
      }
    }
  }
  // combine the pixel-parity splits through LDS (fixed order), then write the slab
  __syncthreads();
  float* s_red = tsm;  // [PSPLIT][NT][KK*4]
  if (PSPLIT > 1) {
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) s_red[((size_t)ps * NT + (tid % NT)) * (KK * 4) + t * 4 + c] = acc[t][c];
    __syncthreads();
  }
  if (ps == 0) {
    float* out = part + (size_t)blockIdx.x * CI * CO * KK;
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = acc[t][c];
        if (PSPLIT > 1)
          for (int sp = 1; sp < PSPLIT; ++sp) v += s_red[((size_t)sp * NT + (tid % NT)) * (KK * 4) + t * 4 + c];
        out[((size_t)ci * CO + co0 + q * 4 + c) * KK + t] = v;
      }
  }
}


// ---------------------------------------------------------------------------------------------
// decnn.10  ConvTranspose2d(16 -> 1, k5, s1, p2): one output channel, so the register block runs over
// PIXELS instead of channels.  T1b: a thread owns 4 consecutive output pixels of a row; per (ci, ky) it reads
// 8 consecutive inputs (2 x ds_read_b128) and the 5 taps of that kernel row (broadcast) for 20 FMAs.
// ---------------------------------------------------------------------------------------------
template <int IPB>
__global__ __launch_bounds__(256) void k_dec10_fwd(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ y, int B) {
  constexpr int CI = 16, H = 28, WP = 32, IMG = CI * WP * WP;   // stored index = i + 2
  float* s_img = tsm;                  // [IPB][CI][32][32]
  float* s_w = tsm + IPB * IMG;        // [CI][5][8]  (kx 0..4, padded to 8)
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * IPB;
  const int nimg = min(IPB, B - b0);
  for (int e = tid; e < IPB * IMG; e += 256) {
    const int im = e / IMG, r = e % IMG, ci = r / (WP * WP), yy = (r / WP) % WP - 2, xx = r % WP - 2;
    float v = 0.f;
    if (im < nimg && yy >= 0 && yy < H && xx >= 0 && xx < H) v = x[(((size_t)(b0 + im) * CI + ci) * H + yy) * H + xx];
    s_img[e] = v;
  }
  for (int e = tid; e < CI * 5 * 8; e += 256) {
    const int kx = e % 8, ky = (e / 8) % 5, ci = e / 40;
    s_w[e] = kx < 5 ? w[(size_t)ci * 25 + ky * 5 + kx] : 0.f;
  }
  __syncthreads();
  const float bv = bias ? bias[0] : 0.f;
  const int items = nimg * H * 7;
  for (int it = tid; it < items; it += 256) {
    const int seg = it % 7, oy = (it / 7) % H, im = it / (7 * H);
    const int ox0 = seg * 4;
    float acc[4] = {bv, bv, bv, bv};
    const float* img = s_img + im * IMG;
#pragma unroll 2
    for (int ci = 0; ci < CI; ++ci) {
#pragma unroll
      for (int ky = 0; ky < 5; ++ky) {
        // iy = oy + 2 - ky -> stored row oy + 4 - ky ; stored col of (pixel j, tap kx) = ox0 + j + 4 - kx
        const float4* rp = reinterpret_cast<const float4*>(img + (ci * WP + (oy + 4 - ky)) * WP + ox0);
        const float4 a = rp[0], b = rp[1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        const float4 w4 = *reinterpret_cast<const float4*>(s_w + (ci * 5 + ky) * 8);
        const float w5 = s_w[(ci * 5 + ky) * 8 + 4];
        const float wk[5] = {w4.x, w4.y, w4.z, w4.w, w5};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int kx = 0; kx < 5; ++kx) acc[j] = fmaf(v[j + 4 - kx], wk[kx], acc[j]);
      }
    }
    *reinterpret_cast<float4*>(y + ((size_t)(b0 + im) * H + oy) * H + ox0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  }
}

// T3b: gw[ci,0,ky,kx] = sum_{b,iy,ix} x[b,ci,iy,ix] gy[b,0,iy-2+ky,ix-2+kx].  thread = (ci, 1 of 16 pixel lanes).
__global__ __launch_bounds__(256) void k_dec10_wgrad(const float* __restrict__ x, const float* __restrict__ gy,
                                                      float* __restrict__ part, int B, int b_per_split) {
  constexpr int CI = 16, H = 28, NP = H * H, XS = NP + 1, GW = 32;   // x rows padded to 785 floats (bank spread)
  float* s_x = tsm;                    // [CI][785]
  float* s_g = tsm + CI * XS;          // [32][32], index = o + 2
  const int tid = threadIdx.x, ci = tid & 15, pl = tid >> 4;
  const int b0 = blockIdx.x * b_per_split, b1 = min(B, b0 + b_per_split);
  float acc[25];
#pragma unroll
  for (int t = 0; t < 25; ++t) acc[t] = 0.f;
  for (int b = b0; b < b1; ++b) {
    __syncthreads();
    for (int e = tid; e < CI * NP; e += 256) s_x[(e / NP) * XS + e % NP] = x[(size_t)b * CI * NP + e];
    for (int e = tid; e < GW * GW; e += 256) {
      const int yy = e / GW - 2, xx = e % GW - 2;
      s_g[e] = (yy >= 0 && yy < H && xx >= 0 && xx < H) ? gy[(size_t)b * NP + yy * H + xx] : 0.f;
    }
    __syncthreads();
    for (int p = pl; p < NP; p += 16) {
      const int iy = p / H, ix = p % H;
      const float xv = s_x[ci * XS + p];
      const float* gp = s_g + iy * GW + ix;   // (oy + 2) = iy + ky
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) acc[ky * 5 + kx] = fmaf(xv, gp[ky * GW + kx], acc[ky * 5 + kx]);
    }
  }
  __syncthreads();
  float* s_red = tsm;  // [16 pixel lanes][16 ci][25]
#pragma unroll
  for (int t = 0; t < 25; ++t) s_red[(pl * 16 + ci) * 25 + t] = acc[t];
  __syncthreads();
  for (int e = tid; e < CI * 25; e += 256) {
    float v = 0.f;
    for (int l = 0; l < 16; ++l) v += s_red[l * 400 + e];
    part[(size_t)blockIdx.x * 400 + e] = v;
  }
}

__global__ void k_sum_splits_t(const float* __restrict__ part, int nsplit, size_t n, float* __restrict__ out) {
  const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float acc = 0.f;
  for (int s = 0; s < nsplit; ++s) acc += part[(size_t)s * n + e];
  out[e] = acc;
}

// ---------------------------------------------------------------------------------------------
// host: dispatch by geometry; return -1 when no tiled specialisation applies (caller falls back to the
// generic direct kernels of vae_conv.hip)
// ---------------------------------------------------------------------------------------------
template <class L> static bool matches(int Ci_conv, int Co_conv, int H, int Ho, int K, int S, int P) {
  // conv geometry of the adjoint: "input" (B, Ci_conv = L::CO, H = L::HO), "output" (B, Co_conv = L::CI, Ho = L::HI)
  return Ci_conv == L::CO && Co_conv == L::CI && H == L::HO && Ho == L::HI && K == L::K && S == L::S && P == L::P;
}

static bool use_mfma() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("GPODE_CONV_VALU"); v = (e && e[0] == '1') ? 0 : 1; }
  return v == 1;
}

// matrix-core path (conv_mfma.hpp): persistent grid of one 512-thread workgroup per CU
// NTHR = 256 with a footprint <= 80 KB: TWO independent 4-wavefront workgroups per CU, whose scatter / store phases interleave
// with each other's MFMA phases instead of idling the matrix pipe in lockstep
// workgroups a BnSink's partial sums must have room for (every statistics-producing launch below stays within it)
constexpr int kSinkMaxWg = 512;

template <class PL, int IPB, int TG, int NCJ, bool PAIR = false, int NTHR = 512, bool DB = false, bool PC = false, bool STATS = false>
static int launch_igemm(const float* x, const float* w, const float* bias, float* y, int B, hipStream_t st, const char* what,
                        const float* in_bn = nullptr, const BnSink* sink = nullptr) {
  constexpr size_t lds = igemm_lds_bytes<PL, IPB, DB, STATS>();
  static_assert(lds <= (NTHR >= 512 ? 160 : 80) * 1024, "LDS budget");
  auto km = k_conv_igemm<PL, IPB, TG, NCJ, PAIR, NTHR, DB, PC, STATS>;
  if (set_max_lds((const void*)km, lds)) return 1;
  const int ngroups = (B + IPB - 1) / IPB;
  const int cap = num_cus() * (NTHR >= 512 ? 1 : 2);
  if (STATS && cap > kSinkMaxWg) return set_error("%s: more workgroups than the statistics scratch holds", what);
  hipLaunchKernelGGL(km, ngroups < cap ? ngroups : cap, NTHR, lds, st, x, w, bias, y, B, in_bn, STATS ? *sink : BnSink{});
  return check_launch(what);
}

static bool conv_pc_enabled() {     // A/B only (GPODE_CONV_PC=1): see launch_T1
  static const bool on = [] { const char* e = getenv("GPODE_CONV_PC"); return e && e[0] == '1'; }();
  return on;
}

// IPB: images per workgroup of the VALU kernel; IPBM / COS: images per group and output channels per pass of the MFMA kernel
template <class L, int IPB, int IPBM, int COS>
static int launch_T1(const float* x, const float* w, const float* bias, float* y, int B, hipStream_t st, const float* in_bn,
                     const BnSink* sink = nullptr) {
  constexpr int MAXTAPS = ((L::K + L::S - 1) / L::S) * ((L::K + L::S - 1) / L::S);
  const size_t lds = sizeof(float) * ((size_t)IPB * L::CI * L::HP * L::HP + (size_t)MAXTAPS * L::CI * L::CO);
  if (use_mfma() && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    constexpr int TG = COS >= 64 ? 1 : (COS >= 32 ? 2 : 4);
    // column-parity classes pair up when both cover the same pixel grid: stride 2, (HO + P) even
    constexpr bool PAIR = L::S == 2 && (L::HO % 2 == 0) && (L::P % 2 == 1);
    if constexpr (std::is_same<L, Dec7>::value) {
      // GPODE_DEC7_FWD_DB=1 (A/B only): ONE image per group in double-buffered planes (2 x 38 KB + 50 KB of slabs), one barrier
      // per group, the two wavefronts of a SIMD scattering at opposite ends of it.  Measured SLOWER than the default below
      // (4096 images: 0.244 vs 0.229 ms): a single image per group quantises to 13 tiles for 12.25 and the scatter was 2 % of
      // the kernel to begin with -- the idle matrix-pipe cycles are not in the phases the second buffer overlaps.
      static const bool db = [] { const char* e = getenv("GPODE_DEC7_FWD_DB"); return e && e[0] == '1'; }();
      if (db) return launch_igemm<FwdPolicy<L, COS>, 1, TG, COS / 16, PAIR, 512, true>(x, w, bias, y, B, st, "convT_fwd_mfma", in_bn);
      // GPODE_CONV_PC=1 (A/B only): producer / consumer wavefronts (conv_mfma.hpp, PC) -- one image per buffer, 8 multiplying + 4
      // fetching wavefronts, one barrier per image.  Measured SLOWER as a port of this engine's job loop (4096 images: 0.274 vs
      // 0.241 ms): three wavefronts per SIMD leave 168 registers, which the tile-group loop (16 copies of the k loop after the
      // compiler's unswitching) only fits with one pixel tile per job -- two LDS operand reads per MFMA and conditional prefetches.
      // The weight-gradient engine (conv_wgrad_v2.hpp) got its gain from a loop written FOR that budget; this one would need the same.
      if (conv_pc_enabled() && B > 2 * num_cus() && !sink)
        return launch_igemm<FwdPolicy<L, COS>, 1, 2, COS / 16, PAIR, 768, true, true>(x, w, bias, y, B, st, "convT_fwd_mfma_pc", in_bn);
    }
    if (sink) return launch_igemm<FwdPolicy<L, COS>, IPBM, TG, COS / 16, PAIR, 512, false, false, true>(x, w, bias, y, B, st, "convT_fwd_mfma_stats", in_bn, sink);
    return launch_igemm<FwdPolicy<L, COS>, IPBM, TG, COS / 16, PAIR>(x, w, bias, y, B, st, "convT_fwd_mfma", in_bn);
  }
  if (sink) return set_error("convT forward with output statistics needs the matrix-core path (16-byte aligned input, GPODE_CONV_VALU unset)");
  if (in_bn) return set_error("convT forward with a fused BatchNorm input needs the matrix-core path (16-byte aligned input, GPODE_CONV_VALU unset)");
  auto kern = k_convT_fwd<L, IPB>;
  if (set_max_lds((const void*)kern, lds)) return 1;
  hipLaunchKernelGGL(kern, (B + IPB - 1) / IPB, 256, lds, st, x, w, bias, y, B);
  return check_launch("convT_fwd_tiled");
}

// IPBM / CIS / TG / NCJ: images per group, input channels per pass, pixel tiles and channel tiles per job of the MFMA kernel
template <class L, int IPB, int COC, int IPBM, int CIS, int TG, int NCJ, int NTHR = 512>
static int launch_T2(const float* gy, const float* w, float* gx, int B, hipStream_t st) {
  if constexpr (IPBM > 0) {
    if (use_mfma() && (reinterpret_cast<uintptr_t>(gy) & 15) == 0)
      return launch_igemm<BwdDataPolicy<L, CIS>, IPBM, TG, NCJ, false, NTHR>(gy, w, nullptr, gx, B, st, "convT_bwd_data_mfma");
  }
  const size_t lds = sizeof(float) * ((size_t)IPB * L::CO * L::GP_ * L::GP_ + (size_t)COC * L::K * L::K * L::CI);
  auto kern = k_convT_bwd_data<L, IPB, COC>;
  if (set_max_lds((const void*)kern, lds)) return 1;
  hipLaunchKernelGGL(kern, (B + IPB - 1) / IPB, 256, lds, st, gy, w, gx, B);
  return check_launch("convT_bwd_data_tiled");
}

template <class L, int IPBM, int WM, int WN, int WT, bool PIPE>
static int launch_wgrad_mfma(const float* x, const float* gy, float* gw, float* scratch, int B, hipStream_t st, const float* in_bn) {
  constexpr int NTHR = WM * WN * WT * 64;            // 512: one workgroup per CU; 256: two (footprint <= 80 KB)
  constexpr size_t ldsm = wgrad_lds_bytes<L, IPBM>();
  static_assert(ldsm <= (NTHR == 512 ? 160 : 80) * 1024, "LDS budget");
  auto km = k_convT_wgrad_mfma<L, IPBM, WM, WN, WT, PIPE, false, NTHR>;
  auto kb = k_convT_wgrad_mfma<L, IPBM, WM, WN, WT, PIPE, true, NTHR>;
  if (set_max_lds((const void*)km, ldsm) || set_max_lds((const void*)kb, ldsm)) return 1;
  const int ngroups = (B + IPBM - 1) / IPBM;
  const int cap = num_cus() * (512 / NTHR);
  const int nwg = ngroups < cap ? ngroups : cap;
  if (in_bn) hipLaunchKernelGGL(kb, nwg, NTHR, ldsm, st, x, gy, scratch, B, in_bn);
  else hipLaunchKernelGGL(km, nwg, NTHR, ldsm, st, x, gy, scratch, B, in_bn);
  const size_t n = (size_t)L::CI * L::CO * L::K * L::K;
  if (reduce_job(RedJob{scratch, gw, nwg, (int)n, 1, L::CI / 16, L::CO / 16, L::K * L::K}, st)) return 1;
  return check_launch("convT_wgrad_mfma");
}

// second engine (conv_wgrad_v2.hpp, its own translation unit vae_wgrad_v2.hip): 8 consumer + 4 producer wavefronts, one image per plane
// buffer.  GPODE_WGRAD_V1=1: the first engine (A/B)
int wgrad_v2_dec7(const float* x, const float* gy, float* gw, float* scratch, int B, hipStream_t st, const float* in_bn);
int wgrad_v2_dec4(const float* x, const float* gy, float* gw, float* scratch, int B, hipStream_t st, const float* in_bn);
static bool wgrad_v2_enabled() {
  static const bool off = [] { const char* e = getenv("GPODE_WGRAD_V1"); return e && e[0] == '1'; }();
  return !off && use_mfma();
}

// IPBM / WM x WN x WT: images per group and wavefront split (ci tiles, co tiles, taps) of the MFMA kernel
template <class L, int COW, int IPBM, int WM, int WN, int WT, bool PIPE>
static int launch_T3(const float* x, const float* gy, float* gw, float* scratch, int B, hipStream_t st, const float* in_bn) {
  if (use_mfma() && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0)
    return launch_wgrad_mfma<L, IPBM, WM, WN, WT, PIPE>(x, gy, gw, scratch, B, st, in_bn);
  if (in_bn) return set_error("convT weight gradient with a fused BatchNorm input needs the matrix-core path");
  constexpr int NT = L::CI * (COW / 4), PSPLIT = 256 / NT, KK = L::K * L::K;
  size_t fl = (size_t)L::HI * L::HI * L::CI + (size_t)L::GP_ * L::GP_ * COW;
  const size_t red = PSPLIT > 1 ? (size_t)PSPLIT * NT * KK * 4 : 0;
  if (red > fl) fl = red;
  const size_t lds = sizeof(float) * fl;
  int nsplit = B < 256 ? B : 256;
  const int bps = (B + nsplit - 1) / nsplit;
  nsplit = (B + bps - 1) / bps;
  auto kern = k_convT_wgrad<L, COW>;
  if (set_max_lds((const void*)kern, lds)) return 1;
  hipLaunchKernelGGL(kern, dim3(nsplit, L::CO / COW), 256, lds, st, x, gy, scratch, B, bps);
  const size_t n = (size_t)L::CI * L::CO * KK;
  if (reduce_job(RedJob{scratch, gw, nsplit, (int)n, 0, 0, 0, 0}, st)) return 1;
  return check_launch("convT_wgrad_tiled");
}


// ConvTranspose2d forward (called with the conv geometry of its adjoint, as conv2d_bwd_data is)
int tiled_bwd_data(const float* gy, const float* w, const float* bias, float* gx, int B, int Ci, int H, int W, int Co, int K, int S,
                   int P, int Ho, int Wo, const float* in_bn, hipStream_t st, const BnSink* sink) {
  if (H != W || Ho != Wo) return -1;
  if (sink && !(use_mfma() && (matches<Dec7>(Ci, Co, H, Ho, K, S, P) || matches<Dec4>(Ci, Co, H, Ho, K, S, P) ||
                               (matches<Dec1>(Ci, Co, H, Ho, K, S, P) && !in_bn))))
    return set_error("gpode_convT_fwd_stats: no specialisation for this geometry");
  if (matches<Dec7>(Ci, Co, H, Ho, K, S, P)) return launch_T1<Dec7, 3, 2, 16>(gy, w, bias, gx, B, st, in_bn, sink);
  if (matches<Dec4>(Ci, Co, H, Ho, K, S, P)) {
    // taps-as-columns form (conv_dec4_mfma.hpp): 48.5 vs 59.7 us for the stage at 512 images (no weight slabs to stage before the first
    // image), 276 vs 267 us at 4096 -- so it takes the small batches (configs[0]: 512 images); GPODE_DEC4_TAPCOLS=0 / 1 forces one
    static const int tapcols = [] { const char* e = getenv("GPODE_DEC4_TAPCOLS"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    if ((tapcols == 1 || (tapcols < 0 && B <= 4 * num_cus())) && use_mfma()) {
      const size_t lds = sizeof(float) * (dec4::NPI * dec4::TLD + 4 * dec4::CI + (sink ? 2 * dec4::CO * dec4::NPO : 0));
      if (set_max_lds((const void*)dec4::k_fwd<true>, lds) || set_max_lds((const void*)dec4::k_fwd<false>, lds) ||
          set_max_lds((const void*)dec4::k_fwd<true, true>, lds) || set_max_lds((const void*)dec4::k_fwd<false, true>, lds)) return 1;
      const int nwg = B < num_cus() ? B : num_cus();
      const BnSink sk = sink ? *sink : BnSink{};
      if (sink && in_bn) hipLaunchKernelGGL((dec4::k_fwd<true, true>), nwg, 512, lds, st, gy, w, bias, gx, B, in_bn, sk);
      else if (sink) hipLaunchKernelGGL((dec4::k_fwd<false, true>), nwg, 512, lds, st, gy, w, bias, gx, B, in_bn, sk);
      else if (in_bn) hipLaunchKernelGGL((dec4::k_fwd<true>), nwg, 512, lds, st, gy, w, bias, gx, B, in_bn, sk);
      else hipLaunchKernelGGL((dec4::k_fwd<false>), nwg, 512, lds, st, gy, w, bias, gx, B, in_bn, sk);
      return check_launch("dec4_fwd_tapcols");
    }
    return launch_T1<Dec4, 3, 2, 16>(gy, w, bias, gx, B, st, in_bn, sink);
  }
  if (matches<Dec1>(Ci, Co, H, Ho, K, S, P)) {
    static const bool old = [] { const char* e = getenv("GPODE_DEC1_ENGINE"); return e && e[0] == '1'; }();
    if (!in_bn && use_mfma() && (!old || sink)) {    // taps folded into the GEMM's columns, weights resident in registers
      const size_t lds = sizeof(float) * 2 * dec1::NPI * dec1::TLD;
      if (set_max_lds((const void*)dec1::k_fwd<false>, lds) || set_max_lds((const void*)dec1::k_fwd<true>, lds)) return 1;
      const int cap = 2 * num_cus();                 // 74 KB of LDS per workgroup
      if (sink) hipLaunchKernelGGL(dec1::k_fwd<true>, B < cap ? B : cap, 256, lds, st, gy, w, bias, gx, B, *sink);
      else hipLaunchKernelGGL(dec1::k_fwd<false>, B < cap ? B : cap, 256, lds, st, gy, w, bias, gx, B, BnSink{});
      return check_launch("dec1_fwd_mfma");
    }
    return launch_T1<Dec1, 8, 8, 64>(gy, w, bias, gx, B, st, in_bn);
  }
  if (matches<Enc3>(Ci, Co, H, Ho, K, S, P) && !in_bn && B >= 96) {
    // d/d input of the encoder's cnn.3 (16 -> 8 channels: no matrix-core tile shape) from 96 images on: the LDS-tiled vector kernel, two
    // images per workgroup -- the direct kernel re-reads every gradient value from L2 for each of its taps (256 images: 38 us direct,
    // 24.5 us tiled with four images per workgroup; 32 images: 21 us direct, 27 us tiled -- 8 workgroups)
    constexpr int IPB = 2;
    constexpr int MAXTAPS = ((Enc3::K + Enc3::S - 1) / Enc3::S) * ((Enc3::K + Enc3::S - 1) / Enc3::S);
    const size_t lds = sizeof(float) * ((size_t)IPB * Enc3::CI * Enc3::HP * Enc3::HP + (size_t)MAXTAPS * Enc3::CI * Enc3::CO);
    auto kern = k_convT_fwd<Enc3, IPB>;
    if (set_max_lds((const void*)kern, lds)) return 1;
    hipLaunchKernelGGL(kern, (B + IPB - 1) / IPB, 256, lds, st, gy, w, bias, gx, B);
    return check_launch("enc_conv3_bwd_data_tiled");
  }
  if (matches<Enc6>(Ci, Co, H, Ho, K, S, P) && use_mfma() && (reinterpret_cast<uintptr_t>(gy) & 15) == 0)   // d/d input of the encoder's cnn.6
  {
    // 8 images per group fill the chip from 2048 images on; a minibatch of 256 made 32 workgroups (73 us on an eighth of the CUs):
    // 2 images per group below half a wave of groups (GPODE_ENC6_IPB8=1: the former launch, A/B)
    static const bool ipb8 = [] { const char* e = getenv("GPODE_ENC6_IPB8"); return e && e[0] == '1'; }();
    if (!ipb8 && B < 4 * num_cus()) return launch_igemm<FwdPolicy<Enc6, 16>, 2, 4, 1>(gy, w, bias, gx, B, st, "enc_conv6_bwd_data_mfma", in_bn);
    return launch_igemm<FwdPolicy<Enc6, 16>, 8, 4, 1>(gy, w, bias, gx, B, st, "enc_conv6_bwd_data_mfma", in_bn);
  }
  if (matches<Dec10>(Ci, Co, H, Ho, K, S, P)) {
    if (use_mfma()) {
      const size_t ldsm = sizeof(float) * dec10::KK * dec10::PST;
      if (set_max_lds((const void*)dec10::k_fwd<true>, ldsm) || set_max_lds((const void*)dec10::k_fwd<false>, ldsm)) return 1;
      if (in_bn) hipLaunchKernelGGL(dec10::k_fwd<true>, B < num_cus() ? B : num_cus(), 512, ldsm, st, gy, w, bias, gx, B, in_bn);
      else hipLaunchKernelGGL(dec10::k_fwd<false>, B < num_cus() ? B : num_cus(), 512, ldsm, st, gy, w, bias, gx, B, in_bn);
      return check_launch("dec10_fwd_mfma");
    }
    constexpr int IPB = 2;
    const size_t lds = sizeof(float) * ((size_t)IPB * 16 * 32 * 32 + 16 * 5 * 8);
    auto kern = k_dec10_fwd<IPB>;
    if (set_max_lds((const void*)kern, lds)) return 1;
    hipLaunchKernelGGL(kern, (B + IPB - 1) / IPB, 256, lds, st, gy, w, bias, gx, B);
    return check_launch("dec10_fwd");
  }
  return -1;
}

// ConvTranspose2d d/d input (conv geometry: x := grad_output (B,Ci,H,W) -> y (B,Co,Ho,Wo)), no bias
int tiled_fwd(const float* x, const float* w, const float* bias, float* y, int B, int Ci, int H, int W, int Co, int K, int S, int P,
              int Ho, int Wo, hipStream_t st) {
  if (H != W || Ho != Wo) return -1;
  if (use_mfma() && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {   // encoder Conv2d layers (with their bias)
    if (matches<Enc3>(Ci, Co, H, Ho, K, S, P)) return launch_igemm<BwdDataPolicy<Enc3, 16>, 4, 4, 1>(x, w, bias, y, B, st, "enc_conv3_fwd_mfma");
    if (matches<Enc6>(Ci, Co, H, Ho, K, S, P)) return launch_igemm<BwdDataPolicy<Enc6, 32>, 8, 1, 2>(x, w, bias, y, B, st, "enc_conv6_fwd_mfma");
  }
  if (bias) return -1;
  if (matches<Dec7>(Ci, Co, H, Ho, K, S, P)) return launch_T2<Dec7, 2, 8, 1, 16, 2, 1, 256>(x, w, y, B, st);
  // decnn.4 d/d input: 36 output pixels per image; two images and two 16-channel halves per pass make 10 equal jobs for 8
  // wavefronts (tools/convt_probe.hip: 28 % of wavefront 0's cycles in the group barrier).  Three images per group with 16
  // channels per pass (7 jobs, one idle wavefront) was measured and is NOT faster (0.455 vs 0.444 ms for d/d input + d/d weight at
  // 4096 images: twice the passes over the source images eat the gain); GPODE_DEC4_BWD_3IMG=1 selects it for A/B.
  if (matches<Dec4>(Ci, Co, H, Ho, K, S, P)) {
    static const bool alt = [] { const char* e = getenv("GPODE_DEC4_BWD_3IMG"); return e && e[0] == '1'; }();
    if (alt) return launch_T2<Dec4, 3, 8, 3, 16, 1, 1>(x, w, y, B, st);
    return launch_T2<Dec4, 3, 8, 2, 32, 1, 1>(x, w, y, B, st);
  }
  if (matches<Dec1>(Ci, Co, H, Ho, K, S, P)) {
    static const bool old = [] { const char* e = getenv("GPODE_DEC1_ENGINE"); return e && e[0] == '1'; }();
    if (use_mfma() && !old && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
      const size_t lds = sizeof(float) * (2 * dec1::GYF + 4 * dec1::NPI * 33);
      const int cap = 2 * num_cus();
      hipLaunchKernelGGL(dec1::k_bwd_data, B < cap ? B : cap, 256, lds, st, x, w, y, B);
      return check_launch("dec1_bwd_data_mfma");
    }
    return launch_T2<Dec1, 8, 32, 6, 32, 1, 1>(x, w, y, B, st);
  }

  if (matches<Dec10>(Ci, Co, H, Ho, K, S, P)) {
    if (use_mfma() && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
      constexpr int IPB = 8;
      const size_t ldsm = sizeof(float) * IPB * dec10::PLANE;
      const int ngroups = (B + IPB - 1) / IPB;
      hipLaunchKernelGGL(dec10::k_bwd_data<IPB>, ngroups < num_cus() ? ngroups : num_cus(), 512, ldsm, st, x, w, y, B);
      return check_launch("dec10_bwd_data_mfma");
    }
    return launch_T2<Dec10, 1, 1, 0, 16, 1, 1>(x, w, y, B, st);
  }
  return -1;
}

// decnn.10's input gradient fused with the backward of the BatchNorm + ReLU in front of it (conv_dec10_mfma.hpp, k_bwd_data_bn).
// scratch: part[nwg][16][2] | part_gx[nwg][16][2], nwg <= kDec10BnMaxWg.  The grid is a function of B alone, so the two
// passes (and the partials they exchange through scratch) agree.
constexpr int kDec10BnMaxWg = 512;
int dec10_bn_scratch_floats() { return 2 * kDec10BnMaxWg * dec10::CI * 2; }

__global__ __launch_bounds__(512) void k_dec10_parts_reduce(const float* __restrict__ part, int nsplit, float* __restrict__ out, int stride2) {
  __shared__ float sA[dec10::CI], sB[dec10::CI];
  dec10::reduce_parts(part, nsplit, sA, sB);
  __syncthreads();
  const int tid = threadIdx.x;
  if (stride2) { if (tid < 2 * dec10::CI) out[tid] = (tid & 1 ? sB : sA)[tid >> 1]; }   // {sum g, sum g xhat} interleaved
  else if (tid < dec10::CI) out[tid] = sA[tid];
}

template <int IPB, int MODE, bool WGRAD = false, typename... A>
static int launch_dec10_bn(int B, hipStream_t st, A... args) {
  const size_t lds = sizeof(float) * ((size_t)IPB * dec10::PLANE + 8 * 4 * dec10::CI * 2 + 2 * dec10::CI + (WGRAD ? 8 * dec10::CI * 32 : 0));
  auto kern = dec10::k_bwd_data_bn<IPB, MODE, WGRAD>;
  if (set_max_lds((const void*)kern, lds)) return 1;
  const int ngroups = (B + IPB - 1) / IPB, cap = 2 * num_cus() < kDec10BnMaxWg ? 2 * num_cus() : kDec10BnMaxWg;
  hipLaunchKernelGGL(kern, ngroups < cap ? ngroups : cap, 512, lds, st, args...);
  return 0;
}
static int dec10_bn_nwg(int B, int& ipb) {
  ipb = B >= 4 * num_cus() ? 4 : 2;
  const int ngroups = (B + ipb - 1) / ipb, cap = 2 * num_cus() < kDec10BnMaxWg ? 2 * num_cus() : kDec10BnMaxWg;
  return ngroups < cap ? ngroups : cap;
}

// gw != nullptr: the layer's weight gradient (16 x 25) is produced by the same pass (wscratch: dec10_bn_wgrad_scratch_floats() floats)
int dec10_bn_wgrad_scratch_floats() { return kDec10BnMaxWg * dec10::CI * dec10::KK + kDec10BnMaxWg; }
int dec10_bn_bwd_sums(const float* c, const float* gy, const float* w, const float* gamma, const float* beta, const float* mean,
                      const float* invstd, float* sums, int B, float* scratch, hipStream_t st, float* gw, float* wscratch, float* gbias) {
  if (((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(c)) & 15) != 0) return set_error("gpode_dec10_bn_bwd_sums: c / gy must be 16-byte aligned");
  if (gw && !wscratch) return set_error("gpode_dec10_bn_bwd_sums_wgrad: scratch for the weight-gradient partials missing");
  int ipb;
  const int nwg = dec10_bn_nwg(B, ipb);
  float* np_ = nullptr;
  const float* cnp = nullptr;
  int rc;
  if (gw) {
    float* part_b = gbias ? wscratch + (size_t)kDec10BnMaxWg * dec10::CI * dec10::KK : nullptr;
    if (ipb == 4) rc = launch_dec10_bn<4, 0, true>(B, st, gy, w, c, gamma, beta, mean, invstd, B, scratch, 0, 0.f, cnp, cnp, 0, np_, np_, np_, np_, wscratch, part_b);
    else rc = launch_dec10_bn<2, 0, true>(B, st, gy, w, c, gamma, beta, mean, invstd, B, scratch, 0, 0.f, cnp, cnp, 0, np_, np_, np_, np_, wscratch, part_b);
    if (!rc) {
      const RedJob jobs[2] = {RedJob{wscratch, gw, nwg, dec10::CI * dec10::KK, 0, 0, 0, 0}, RedJob{part_b, gbias, nwg, 1, 0, 0, 0, 0}};
      if (reduce_jobs(jobs, gbias ? 2 : 1, st)) return 1;
    }
  } else if (ipb == 4) rc = launch_dec10_bn<4, 0>(B, st, gy, w, c, gamma, beta, mean, invstd, B, scratch, 0, 0.f, cnp, cnp, 0, np_, np_, np_, np_, np_, np_);
  else rc = launch_dec10_bn<2, 0>(B, st, gy, w, c, gamma, beta, mean, invstd, B, scratch, 0, 0.f, cnp, cnp, 0, np_, np_, np_, np_, np_, np_);
  if (rc) return rc;
  if (sums) hipLaunchKernelGGL(k_dec10_parts_reduce, 1, 512, 0, st, scratch, nwg, sums, 1);
  return check_launch("dec10_bn_bwd_sums");
}

int dec10_bn_bwd_apply(const float* c, const float* gy, const float* w, const float* gamma, const float* beta, const float* mean,
                       const float* invstd, const float* gathered, const float* wts, int W, float count_all, float* gc, float* ggamma,
                       float* gbeta, float* gc_chansum, int B, float* scratch, hipStream_t st) {
  if (((reinterpret_cast<uintptr_t>(gy) | reinterpret_cast<uintptr_t>(c) | reinterpret_cast<uintptr_t>(gc)) & 15) != 0)
    return set_error("gpode_dec10_bn_bwd_apply: c / gy / gc must be 16-byte aligned");
  if (gathered && (W < 1 || !wts)) return set_error("gpode_dec10_bn_bwd_apply: gathered sums need their weights");
  int ipb;
  const int nwg = dec10_bn_nwg(B, ipb);
  float* part_gx = gc_chansum ? scratch + (size_t)kDec10BnMaxWg * dec10::CI * 2 : nullptr;
  const float count = gathered ? count_all : (float)B * dec10::NP;
  int rc;
  if (ipb == 4) rc = launch_dec10_bn<4, 1>(B, st, gy, w, c, gamma, beta, mean, invstd, B, scratch, nwg, count, gathered, wts, W, ggamma, gbeta, gc, part_gx, (float*)nullptr, (float*)nullptr);
  else rc = launch_dec10_bn<2, 1>(B, st, gy, w, c, gamma, beta, mean, invstd, B, scratch, nwg, count, gathered, wts, W, ggamma, gbeta, gc, part_gx, (float*)nullptr, (float*)nullptr);
  if (rc) return rc;
  if (gc_chansum && reduce_job(RedJob{part_gx, gc_chansum, nwg, dec10::CI, 2, 0, 0, 0}, st)) return 1;
  return check_launch("dec10_bn_bwd_apply");
}

// ConvTranspose2d d/d weight (conv geometry: x := grad_output, gy := the layer's input)
int tiled_bwd_weight(const float* x, const float* gy, float* gw, float* scratch, int B, int Ci, int H, int W, int Co, int K, int S,
                     int P, int Ho, int Wo, const float* in_bn, hipStream_t st) {
  if (H != W || Ho != Wo) return -1;
  const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0;
  if (matches<Dec7>(Ci, Co, H, Ho, K, S, P)) {
    if (wgrad_v2_enabled() && aligned) return wgrad_v2_dec7(gy, x, gw, scratch, B, st, in_bn);
    return launch_T3<Dec7, 16, 2, 1, 1, 8, true>(gy, x, gw, scratch, B, st, in_bn);
  }
  if (matches<Dec4>(Ci, Co, H, Ho, K, S, P)) {
    if (wgrad_v2_enabled() && aligned) return wgrad_v2_dec4(gy, x, gw, scratch, B, st, in_bn);
    return launch_T3<Dec4, 16, 2, 4, 2, 1, false>(gy, x, gw, scratch, B, st, in_bn);
  }
  if (matches<Dec1>(Ci, Co, H, Ho, K, S, P)) {
    static const bool old = [] { const char* e = getenv("GPODE_DEC1_ENGINE"); return e && e[0] == '1'; }();
    if (!in_bn && use_mfma() && !old && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {   // gy: the layer's input, x: grad_output
      const size_t lds = sizeof(float) * 2 * dec1::GYF;
      static const int wcap = [] { const char* e = getenv("GPODE_DEC1_WGRAD_WGS"); return e ? atoi(e) : 256; }();   // 256: 33.8 + 7.8 us (kernel + reduction of the partials) at 4096 images; 512: 33.1 + 10.8; 128: 56 + 6
      const int cap = 2 * num_cus() < wcap ? 2 * num_cus() : wcap;
      const int nwg = B < cap ? B : cap;
      hipLaunchKernelGGL(dec1::k_wgrad, nwg, 256, lds, st, gy, x, scratch, B);
      if (reduce_job(RedJob{scratch, gw, nwg, dec1::CI * dec1::NN, 0, 0, 0, 0}, st)) return 1;
      return check_launch("dec1_wgrad_mfma");
    }
    return launch_T3<Dec1, 32, 8, 2, 4, 1, true>(gy, x, gw, scratch, B, st, in_bn);
  }
  if (matches<Enc6>(Ci, Co, H, Ho, K, S, P) && use_mfma() && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0)
    return launch_wgrad_mfma<Enc6, 8, 2, 1, 4, true>(gy, x, gw, scratch, B, st, in_bn);   // d/d weight of the encoder's cnn.6
  if (matches<Dec10>(Ci, Co, H, Ho, K, S, P) && use_mfma() &&
      ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0) {
    constexpr int IPB = 1;
    const size_t fl = (size_t)IPB * dec10::PLANE > 8 * 16 * 32 ? (size_t)IPB * dec10::PLANE : 8 * 16 * 32;
    const int ngroups = (B + IPB - 1) / IPB;
    const int nwg = ngroups < num_cus() ? ngroups : num_cus();
    hipLaunchKernelGGL(dec10::k_wgrad<IPB>, nwg, 512, sizeof(float) * fl, st, gy, x, scratch, B, in_bn);
    if (reduce_job(RedJob{scratch, gw, nwg, 400, 0, 0, 0, 0}, st)) return 1;
    return check_launch("dec10_wgrad_mfma");
  }
  if (matches<Dec10>(Ci, Co, H, Ho, K, S, P)) {
    int nsplit = B < 256 ? B : 256;
    const int bps = (B + nsplit - 1) / nsplit;
    nsplit = (B + bps - 1) / bps;
    const size_t lds = sizeof(float) * (16 * 785 + 32 * 32);   // >= the [16][16][25] reduction buffer
    if (set_max_lds((const void*)k_dec10_wgrad, lds)) return 1;
    hipLaunchKernelGGL(k_dec10_wgrad, nsplit, 256, lds, st, gy, x, scratch, B, bps);
    if (reduce_job(RedJob{scratch, gw, nsplit, 400, 0, 0, 0, 0}, st)) return 1;
    return check_launch("dec10_wgrad");
  }
  return -1;
}

}  // namespace gp

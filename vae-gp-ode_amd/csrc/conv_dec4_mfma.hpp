// conv_dec4_mfma.hpp -- forward of the decoder's second transposed convolution, ConvTranspose2d(64 -> 32, K = 5, stride 2, padding 1)
// from 6 x 6 to 13 x 13 images (reference: experiments/model/core/vae.py:110-111, decnn.4), in the taps-as-columns form of
// conv_dec1_mfma.hpp:
//   T[p][co, tap] = sum_ci a[ci][p] w[ci][co][tap]        (m = 36 input pixels in 3 row tiles, k = 64, n = 32 x 25 = 800)
//   y[co][oy][ox] = b[co] + sum over the <= 9 taps of the output's parity class of T[(iy, ix)][co, tap],  oy = 2 iy - 1 + ky
// a = ReLU(BatchNorm(x)) applied while the A fragments are loaded (in_bn: per-channel {mean, invstd, gamma, beta}).
// 8 wavefronts; wavefront v keeps the B-fragments of column tiles v, v + 8, ... (<= 7 x 16 = 112 VGPRs) for the whole persistent
// loop.  T (36 x 804 floats = 116 KB) lives in LDS, one image at a time.
#pragma once
#include "conv_mfma.hpp"
#include "bn_sink.hpp"

namespace gp {
namespace dec4 {

constexpr int CI = 64, CO = 32, NPI = 36, HI = 6, HO = 13, NPO = 169, KK = 25, NN = CO * KK, NT = NN / 16, KS = CI / 4, NTW = 7;
constexpr int TLD = NN + 4;                          // 804 = 36 mod 64: the four row groups of a tile store 16 banks apart
constexpr int NOUT = (CO * NPO + 511) / 512;         // 11 output slots per thread (5408 outputs)

// STATS: the BatchNorm statistics of the output (decnn.5) are summed while it is stored (bn_sink.hpp)
template <bool HAS_BN, bool STATS = false>
__global__ __launch_bounds__(512, 2) void k_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                float* __restrict__ y, int B, const float* __restrict__ in_bn, BnSink sink) {
  float* T = igemm_smem;                             // [36][TLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int ntw = wave < NT - 8 * (NTW - 1) ? NTW : NTW - 1;          // 50 column tiles over 8 wavefronts: 7 7 6 6 6 6 6 6
  float bw[NTW][KS];
#pragma unroll
  for (int j = 0; j < NTW; ++j)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bw[j][ks] = j < ntw ? w[(size_t)(4 * ks + lk) * NN + 16 * (wave + 8 * j) + lr] : 0.f;
  float4* sTF = reinterpret_cast<float4*>(T + NPI * TLD);             // BatchNorm + ReLU table of the 64 input channels
  if (HAS_BN && tid < CI) sTF[tid] = reinterpret_cast<const float4*>(in_bn)[tid];
  __syncthreads();
  float af[3][KS];
  auto loadA = [&](int b) {                          // A[m = pixel 16 mt + lr][k = ci]; pixels >= 36 of the third tile: clamped, unused
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float v = x[((size_t)b * CI + 4 * ks + lk) * NPI + min(16 * mt + lr, NPI - 1)];
        af[mt][ks] = HAS_BN ? bn_relu(v, sTF[4 * ks + lk]) : v;
      }
  };
  // STATS: running sum (y - k), sum (y - k)^2 of every output slot (each owned by one thread) behind T and the table -- the
  // registers are taken (112 of weight fragments)
  float* S0 = T + NPI * TLD + 4 * CI;
  float* S1 = S0 + CO * NPO;
  if (STATS)
    for (int e = tid; e < 2 * CO * NPO; e += 512) S0[e] = 0.f;
  if ((int)blockIdx.x < B) loadA(blockIdx.x);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
      f32x4 acc[NTW];
#pragma unroll
      for (int j = 0; j < NTW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float a = af[mt][ks];
#pragma unroll
        for (int j = 0; j < NTW; ++j)
          if (j < NTW - 1 || ntw == NTW) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[j][ks], acc[j], 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NTW; ++j)
        if (j < NTW - 1 || ntw == NTW) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int p = 16 * mt + 4 * lk + r;
            if (mt < 2 || p < NPI) T[p * TLD + 16 * (wave + 8 * j) + lr] = acc[j][r];
          }
        }
    }
    __syncthreads();                                 // T of image b complete
    if (b + (int)gridDim.x < B) loadA(b + gridDim.x);                 // in flight under the gather
    float* yb = y + (size_t)b * (CO * NPO);
#pragma unroll 1
    for (int i = 0; i < NOUT; ++i) {
      const int o = tid + 512 * i;
      if (o < CO * NPO) {
        const int co = o / NPO, q = o - co * NPO, oy = q / HO, ox = q - oy * HO;
        const int py = (oy + 1) & 1, px = (ox + 1) & 1, iy0 = (oy + 1 - py) >> 1, ix0 = (ox + 1 - px) >> 1;
        // taps (ky, kx) = (py + 2 a, px + 2 c), source pixel (iy0 - a, ix0 - c)
        const int base = (iy0 * HI + ix0) * TLD + co * KK + py * 5 + px;
        float tv[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) tv[3 * a + c] = T[min(max(base - (a * HI + c) * TLD + 10 * a + 2 * c, 0), NPI * TLD - 1)];
        float v = bias ? bias[co] : 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const bool ok = py + 2 * a < 5 && px + 2 * c < 5 && iy0 - a >= 0 && iy0 - a < HI && ix0 - c >= 0 && ix0 - c < HI;
            v += ok ? tv[3 * a + c] : 0.f;
          }
        yb[o] = v;
        if (STATS) {
          const float d = v - bn_sink_shift(sink, co);
          S0[o] += d;
          S1[o] = fmaf(d, d, S1[o]);
        }
      }
    }
    __syncthreads();                                 // every thread is done with T before the next image overwrites it
  }
  if constexpr (STATS) {
    float* sm = T;                                   // the workgroup's [CO][2]  (T is free: last barrier above)
    if (tid < CO) {
      float a = 0.f, b = 0.f;
      for (int q = 0; q < NPO; ++q) { a += S0[tid * NPO + q]; b += S1[tid * NPO + q]; }
      sm[2 * tid] = a;
      sm[2 * tid + 1] = b;
    }
    __syncthreads();
    bn_sink_publish<CO, 512>(sink, sm);
  }
}

}  // namespace dec4
}  // namespace gp

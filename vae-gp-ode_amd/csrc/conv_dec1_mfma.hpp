// conv_dec1_mfma.hpp -- the decoder's first transposed convolution, ConvTranspose2d(32 -> 64, K = 3, stride 1, padding 0) from
// 4 x 4 to 6 x 6 images (reference: experiments/model/core/vae.py:107-108, decnn.1), on the fp32 matrix cores.
//
// A 4 x 4 image is ONE 16-row MFMA tile, so the layer is a plain GEMM per image with the 9 taps folded into the column dimension,
//   T[p][co, tap] = sum_ci x[ci][p] w[ci][co][tap]          (m = 16 input pixels, k = 32, n = 64 x 9 = 576)
//   y[co][oy][ox] = b[co] + sum_{tap: (oy - ky, ox - kx) inside} T[(oy - ky, ox - kx)][co, tap]        (col2im through LDS)
// which does exactly the layer's 0.295 MMAC per image (the plane-scatter engine of conv_mfma.hpp spent 67 us on 4096 images
// here, 0.22 of the matrix peak: 36-pixel outputs fill its 16-pixel tiles badly and its weight slabs are reloaded per pass).
// The weight operand never leaves the registers: wavefront w of the 4 owns column tiles 9 w .. 9 w + 8 and keeps their
// 9 x 8 B-fragments (72 VGPRs) for the whole persistent loop; A comes straight from global memory (16 consecutive floats per
// channel), one image ahead.  T is double-buffered in LDS, so an image costs one barrier.
#pragma once
#include "conv_mfma.hpp"
#include "bn_sink.hpp"

namespace gp {
namespace dec1 {

constexpr int CI = 32, CO = 64, NPI = 16, HO = 6, NPO = 36, KK = 9, NN = CO * KK, NTW = 9, KS = CI / 4;
constexpr int TLD = NN + 4;                          // row stride of T: 580 = 4 mod 64, the 16 pixel rows start 4 banks apart

// STATS: the BatchNorm statistics of the output (decnn.2) are summed while it is stored (bn_sink.hpp)
template <bool STATS>
__global__ __launch_bounds__(256, 2) void k_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                float* __restrict__ y, int B, BnSink sink) {
  float* sT = igemm_smem;                            // [2][16][TLD]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  // B[k = ci][n = (co, tap)] = w[ci][n]: this wavefront's 9 column tiles, all 8 k-steps
  float bw[NTW][KS];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bw[nt][ks] = w[(size_t)(4 * ks + lk) * NN + 16 * (NTW * wave + nt) + lr];
  // the outputs this thread gathers: o = tid + 256 i < 2304 = 64 x 36 (9 per thread), o = co * 36 + oy * 6 + ox -- the same for
  // every image, so their T offsets, tap masks and biases are worked out once
  constexpr int NOUT = (CO * NPO) / 256;
  int gbase[NOUT];
  unsigned gmask[NOUT];
  float gbias[NOUT];
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    const int o = tid + 256 * i, co = o / NPO, q = o - co * NPO, oy = q / HO, ox = q - oy * HO;
    gbase[i] = (oy * 4 + ox) * TLD + co * KK;        // tap (ky, kx) reads gbase - (4 ky + kx) TLD + 3 ky + kx
    unsigned m = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
        if (oy - ky >= 0 && oy - ky < 4 && ox - kx >= 0 && ox - kx < 4) m |= 1u << (3 * ky + kx);
    gmask[i] = m;
    gbias[i] = bias ? bias[co] : 0.f;
  }
  float gk[NOUT], ss[NOUT], sq[NOUT];                // STATS: shift, sum (y - k), sum (y - k)^2 of this thread's nine outputs
#pragma unroll
  for (int i = 0; i < NOUT; ++i) {
    gk[i] = STATS ? bn_sink_shift(sink, (tid + 256 * i) / NPO) : 0.f;
    ss[i] = sq[i] = 0.f;
  }
  float af[KS];
  auto loadA = [&](int b) {                          // A[m = pixel][k = ci] = x[b][ci][pixel]
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) af[ks] = x[((size_t)b * CI + 4 * ks + lk) * NPI + lr];
  };
  int buf = 0;
  if ((int)blockIdx.x < B) loadA(blockIdx.x);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    float ac[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ac[ks] = af[ks];
    if (b + (int)gridDim.x < B) loadA(b + gridDim.x);
    float* T = sT + buf * NPI * TLD;
    f32x4 acc[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)                  // k outside: nine independent accumulators back to back
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[ks], bw[nt][ks], acc[nt], 0, 0, 0);
    // D[m = pixel 4 lk + r][n = lr of this tile]
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(4 * lk + r) * TLD + 16 * (NTW * wave + nt) + lr] = acc[nt][r];
    __syncthreads();                                 // T of image b complete (the other buffer is free: its readers passed this barrier)
    float* yb = y + (size_t)b * (CO * NPO) + tid;
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
      // all nine taps are read (addresses clamped into the buffer) and the ones outside the image dropped by a select: a
      // branch per tap serialises 81 LDS round trips per image (15k cycles; this form: ~2k)
      float tv[KK];
#pragma unroll
      for (int t = 0; t < KK; ++t) {
        const int a = gbase[i] - (4 * (t / 3) + t % 3) * TLD + t;
        tv[t] = T[min(max(a, 0), NPI * TLD - 1)];
      }
      float v = gbias[i];
#pragma unroll
      for (int t = 0; t < KK; ++t) v += (gmask[i] >> t & 1) ? tv[t] : 0.f;
      yb[256 * i] = v;
      if (STATS) {
        const float d = v - gk[i];
        ss[i] += d;
        sq[i] = fmaf(d, d, sq[i]);
      }
    }
    buf ^= 1;
  }
  if constexpr (STATS) {
    // per-thread sums -> LDS by output index, then thread c adds the 36 entries of channel c in order
    __syncthreads();
    float* S0 = sT;                                  // [2304], [2304], then the workgroup's [CO][2]
    float* S1 = sT + CO * NPO;
    float* sm = sT + 2 * CO * NPO;
#pragma unroll
    for (int i = 0; i < NOUT; ++i) { S0[tid + 256 * i] = ss[i]; S1[tid + 256 * i] = sq[i]; }
    __syncthreads();
    if (tid < CO) {
      float a = 0.f, b = 0.f;
      for (int q = 0; q < NPO; ++q) { a += S0[tid * NPO + q]; b += S1[tid * NPO + q]; }
      sm[2 * tid] = a;
      sm[2 * tid + 1] = b;
    }
    __syncthreads();
    bn_sink_publish<CO, 256>(sink, sm);
  }
}

// ---------------------------------------------------------------------------------------------
// d/d input: gx[ci][p] = sum_{co, tap} gy[co][p + tap] w[ci][co][tap]   (m = 16 input pixels, n = 32, k = 64 x 9 = 576).
// The 4 wavefronts split k by output channel (16 each, all 9 taps: 36 k-steps) and keep their 36 x 2 B-fragments of the weight
// in registers; A is gathered from the image of gy staged in LDS (double-buffered, next image in flight); the four partial
// tiles meet in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int GYF = CO * NPO;                        // 2304 floats per image of gy

__global__ __launch_bounds__(256, 2) void k_bwd_data(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx, int B) {
  float* sG = igemm_smem;                            // [2][2304]
  float* sP = igemm_smem + 2 * GYF;                  // [4][16][32 + 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  // k-step (tap, c4): k = co = 16 wave + 4 c4 + lk at tap;  B[k][n = ci] = w[ci][co][tap]
  float bw[KK][4][2];
#pragma unroll
  for (int t = 0; t < KK; ++t)
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) bw[t][c4][nt] = w[(size_t)(16 * nt + lr) * NN + (16 * wave + 4 * c4 + lk) * KK + t];
  const int abase = (lr >> 2) * HO + (lr & 3);       // window origin of input pixel lr = (iy, ix) in the 6 x 6 plane
  static_assert(GYF / 4 == 576, "three float4 per thread: 256 + 256 + 64");
  float4 p0, p1, p2;                                 // (named, not an array: the array went to scratch memory)
  auto prefetch = [&](int b) {
    const float4* src = reinterpret_cast<const float4*>(gy) + (size_t)b * (GYF / 4);
    p0 = src[tid]; p1 = src[tid + 256]; p2 = src[min(tid + 512, GYF / 4 - 1)];
  };
  auto stage = [&](int buf) {
    float4* dst = reinterpret_cast<float4*>(sG + buf * GYF);
    dst[tid] = p0; dst[tid + 256] = p1;
    if (tid < 64) dst[tid + 512] = p2;
  };
  int buf = 0;
  if ((int)blockIdx.x < B) { prefetch(blockIdx.x); stage(0); }
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const bool more = b + (int)gridDim.x < B;
    if (more) prefetch(b + gridDim.x);
    __syncthreads();                                 // image b staged; partial tiles of the previous image consumed
    const float* G = sG + buf * GYF + abase;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        const float a = G[(16 * wave + 4 * c4 + lk) * NPO + (t / 3) * HO + t % 3];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[t][c4][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[t][c4][1], acc[1], 0, 0, 0);
      }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sP[(wave * NPI + 4 * lk + r) * 33 + 16 * nt + lr] = acc[nt][r];
    if (more) stage(buf ^ 1);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int o = tid + 256 * i, ci = o >> 4, p = o & 15;          // gx[b][ci][p]
      gx[(size_t)b * (CI * NPI) + o] = (sP[(0 * NPI + p) * 33 + ci] + sP[(1 * NPI + p) * 33 + ci]) + (sP[(2 * NPI + p) * 33 + ci] + sP[(3 * NPI + p) * 33 + ci]);
    }
    buf ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------
// d/d weight: gw[ci][co, tap] = sum_b sum_p x[b][ci][p] gy[b][co][p + tap]   (m = 32, n = 576, k = 16 pixels per image).
// Wavefront w owns column tiles 9 w .. 9 w + 8 of both row tiles: 18 accumulator tiles (72 VGPRs) that live through the whole
// persistent loop; k-step ks is input row ks, lane group lk its column.  part[workgroup][ci][co, tap] is reduced by reduce_job.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_wgrad(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part, int B) {
  float* sG = igemm_smem;                            // [2][2304]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  int boff[NTW];                                     // B[k = pixel (ks, lk)][n]: gy[co][(ks + ky) * 6 + lk + kx], n = (co, tap)
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const int n = 16 * (NTW * wave + nt) + lr, co = n / KK, t = n - co * KK;
    boff[nt] = co * NPO + (t / 3) * HO + t % 3 + lk;
  }
  f32x4 acc[2][NTW];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 p0, p1, p2;
  float xa[2][4];                                    // A[m = ci][k = pixel]: x[b][16 mt + lr][4 ks + lk]
  auto prefetch = [&](int b) {
    const float4* src = reinterpret_cast<const float4*>(gy) + (size_t)b * (GYF / 4);
    p0 = src[tid]; p1 = src[tid + 256]; p2 = src[min(tid + 512, GYF / 4 - 1)];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xa[mt][ks] = x[((size_t)b * CI + 16 * mt + lr) * NPI + 4 * ks + lk];
  };
  auto stage = [&](int buf) {
    float4* dst = reinterpret_cast<float4*>(sG + buf * GYF);
    dst[tid] = p0; dst[tid + 256] = p1;
    if (tid < 64) dst[tid + 512] = p2;
  };
  int buf = 0;
  if ((int)blockIdx.x < B) prefetch(blockIdx.x);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    stage(buf);
    float xc[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) xc[mt][ks] = xa[mt][ks];
    if (b + (int)gridDim.x < B) prefetch(b + gridDim.x);
    __syncthreads();                                 // image b staged (the other buffer's readers are past the previous barrier)
    const float* G = sG + buf * GYF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const float bv = G[boff[nt] + ks * HO];
        acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[0][ks], bv, acc[0][nt], 0, 0, 0);
        acc[1][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xc[1][ks], bv, acc[1][nt], 0, 0, 0);
      }
    buf ^= 1;
  }
  // D[m = ci 16 mt + 4 lk + r][n]
  float* pp = part + (size_t)blockIdx.x * (CI * NN);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pp[(size_t)(16 * mt + 4 * lk + r) * NN + 16 * (NTW * wave + nt) + lr] = acc[mt][nt][r];
}

}  // namespace dec1
}  // namespace gp

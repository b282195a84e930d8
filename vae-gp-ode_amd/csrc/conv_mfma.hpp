// conv_mfma.hpp -- implicit-GEMM convolution engine on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32)
// for the decoder's ConvTranspose2d layers (reference: experiments/model/core/vae.py:64-84, decnn.1/4/7) in both
// directions:
//   FwdPolicy      y  = convT(x, w) + bias        (gather form, one GEMM per stride-parity class)
//   BwdDataPolicy  gx = conv(gy, w)               (the adjoint: an ordinary strided convolution)
//
// GEMM view per image group:  D[n][m] = sum_k  W[k][n] * X[k][m]
//   m = output pixels (16 per tile; lanes 0..15 of every 16-lane group),
//   n = output channels (16 per tile),  k = (tap, source channel), 4 source channels per MFMA.
// The product is computed transposed (A := weights, B := pixels), so a lane ends up with one pixel and
// 4 channels per tile and the global stores of a 16-lane group hit neighbouring pixels of one channel plane.
//
// Persistent kernel, one workgroup of NTHR = 512 threads per CU (two wavefronts per SIMD: one wavefront's LDS
// operand waits hide under the other's MFMAs -- tools/mfma_probe.hip measures 32 cycles per MFMA, 58 with a
// dependent LDS fetch and one wavefront, 34 with two):
//   * the weights of all classes for NCS output channels stay resident in LDS ([cls][tap][k][n], staged once
//     per pass; NC / NCS passes over the image groups);
//   * source images live in LDS as zero-padded planes whose stride PS is chosen so that the four 16-lane
//     groups of an operand fetch (4 consecutive k planes x 16 pixels) fall on disjoint banks;
//   * the next group's source (contiguous floats) is fetched as float4 into registers right after the current
//     group was scattered to LDS, so the global latency hides under the current group's MFMAs;
//   * operands of k-step s+1 are fetched into a second register set while the MFMAs of step s issue;
//   * the (class, pixel tile, channel-tile column) jobs of a group are split over the wavefronts by cost
//     (taps), SIMD-wise contiguous.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#include "bn_math.hpp"
#include "bn_sink.hpp"

namespace gp {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// phase timing for tools/convt_probe.hip (compiled out of the library)
#ifdef CONVT_PROBE
__device__ unsigned long long g_probe[8];
#define PROBE_T(v) const long long v = clock64()
#define PROBE_ADD(i, v) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_probe[i] += (unsigned long long)(clock64() - (v)); } while (0)
#else
#define PROBE_T(v)
#define PROBE_ADD(i, v)
#endif

constexpr int cmax_(int a, int b) { return a > b ? a : b; }
// plane stride >= n with stride == 16 (mod 32): unit-stride pixels, 4 planes -> 4 disjoint 16-bank groups per half wave
constexpr int plane_stride_unit(int n) { return ((n + 15) / 32) * 32 + 16; }

// ---------------------------------------------------------------------------------------------
// forward transposed convolution: source = x (CI planes HI x HI), output = y (CO planes HO x HO)
// ---------------------------------------------------------------------------------------------
template <class L, int NCS_> struct FwdPolicy {
  static constexpr int KC = L::CI, NC = L::CO, NCS = NCS_, NCLS = L::S * L::S, KK = L::K * L::K;
  static constexpr int SH = L::HI, OH = L::HO, HP = L::HP, PADL = L::PL;
  static constexpr int PS = plane_stride_unit(HP * HP);
  static constexpr int WROW = NCS;                           // slab row stride [cls][tap][k][n]
  static constexpr int WSLAB = KK * KC * WROW;
  static constexpr int NWE = KC * NCS * KK;                  // weights per pass
  static constexpr int class_tap_offset(int cls) {
    int off = 0;
    for (int c = 0; c < cls; ++c) off += ((L::K - c / L::S + L::S - 1) / L::S) * ((L::K - c % L::S + L::S - 1) / L::S);
    return off;
  }
  // stride-parity class CLS = py * S + px: taps ky = py + S t, output pixels oy = S qy + py - P in [0, HO), source iy = qy - t
  template <int CLS> struct C {
    static constexpr int py = CLS / L::S, px = CLS % L::S;
    static constexpr int nty = (L::K - py + L::S - 1) / L::S, ntx = (L::K - px + L::S - 1) / L::S, ntaps = nty * ntx;
    static constexpr int qy0 = cmax_(0, (L::P - py + L::S - 1) / L::S), qx0 = cmax_(0, (L::P - px + L::S - 1) / L::S);
    static constexpr int ny = (L::HO - 1 + L::P - py) / L::S - qy0 + 1, nx = (L::HO - 1 + L::P - px) / L::S - qx0 + 1;
    static constexpr int npc = ny * nx;                      // pixels of the class per image
    static constexpr int slab_taps = class_tap_offset(CLS);
    static __device__ __forceinline__ int tap_off(int t) { return -(t / ntx) * HP - t % ntx; }
    static __device__ __forceinline__ int pix_addr(int p) { return (qy0 + p / nx + PADL) * HP + qx0 + p % nx + PADL; }
    static constexpr __host__ __device__ int out_off(int p) {
      return (L::S * (qy0 + p / nx) + py - L::P) * OH + L::S * (qx0 + p % nx) + px - L::P;
    }
  };
  // pass-local weight element e (runs of NCS*KK contiguous floats per source channel) -> global index, slab index
  static __device__ __forceinline__ size_t w_src(int e, int n0) {
    const int ci = e / (NCS * KK), r = e % (NCS * KK);
    return ((size_t)ci * NC + n0) * KK + r;
  }
  static __device__ __forceinline__ int w_dst(int e) {
    const int ci = e / (NCS * KK), r = e % (NCS * KK), col = r / KK, ky = (r % KK) / L::K, kx = r % L::K;
    const int py = ky % L::S, px = kx % L::S;
    int off = 0;
    for (int c = 0; c < py * L::S + px; ++c) off += ((L::K - c / L::S + L::S - 1) / L::S) * ((L::K - c % L::S + L::S - 1) / L::S);
    const int tap = (ky / L::S) * ((L::K - px + L::S - 1) / L::S) + kx / L::S;
    return ((off + tap) * KC + ci) * WROW + col;
  }
};

// ---------------------------------------------------------------------------------------------
// d/d input of the transposed convolution: source = gy (CO planes HO x HO, stored at index oy + P),
// output = gx (CI planes HI x HI); gx[ci][iy][ix] = sum_{co,ky,kx} gy[co][S iy - P + ky][S ix - P + kx] w[ci][co][ky][kx]
// ---------------------------------------------------------------------------------------------
template <class L, int NCS_> struct BwdDataPolicy {
  static constexpr int KC = L::CO, NC = L::CI, NCS = NCS_, NCLS = 1, KK = L::K * L::K;
  static constexpr int SH = L::HO, OH = L::HI, HP = L::GP_, PADL = L::P;
  // pixel stride S in the plane: S = 2 wants an odd plane stride (planes k, k+1 then cover the even and the odd banks)
  static constexpr int PS = L::S == 1 ? plane_stride_unit(HP * HP) : (HP * HP) | 1;
  static constexpr int WROW = NCS;                           // [tap][co][ci_local]
  static constexpr int WSLAB = KK * KC * WROW;
  static constexpr int NWE = KC * NCS * KK;
  template <int CLS> struct C {
    static constexpr int ntaps = KK, ntx = L::K, npc = L::HI * L::HI, nx = L::HI, slab_taps = 0;
    static __device__ __forceinline__ int tap_off(int t) { return (t / L::K) * HP + t % L::K; }
    static __device__ __forceinline__ int pix_addr(int p) { return L::S * (p / nx) * HP + L::S * (p % nx); }
    static constexpr __host__ __device__ int out_off(int p) { return p; }
  };
  // w[ci][co][ky][kx]: a pass over NCS input channels is one contiguous block
  static __device__ __forceinline__ size_t w_src(int e, int n0) { return (size_t)n0 * KC * KK + e; }
  static __device__ __forceinline__ int w_dst(int e) {
    const int cil = e / (KC * KK), co = (e / KK) % KC, tap = e % KK;
    return (tap * KC + co) * WROW + cil;
  }
};

// does the per-channel input-transform table fit the unused, 16-byte aligned tail of the channel planes?
template <class PL> constexpr bool igemm_tf_in_pad() { return PL::PS % 4 == 0 && (PL::HP * PL::HP + 3) / 4 * 4 + 4 <= PL::PS; }

// All k-steps (tap x 4*KB source channels) of NG pixel tiles x NCJ channel tiles of one wavefront.
template <class CG, int NG, int TG, int NCJ, int KC, int WROW, int PS, int KBMAX = 8>
__device__ __forceinline__ void igemm_tile_mma(const float* __restrict__ s_img, const float* __restrict__ sw, const int (&abase)[TG],
                                               int lk, int lr, f32x4 (&acc)[TG][NCJ]) {
  constexpr int KB = (KC / 4) < KBMAX ? (KC / 4) : KBMAX;   // MFMA k-steps per fetch batch
  constexpr int NB = KC / (4 * KB);                 // batches per tap
  constexpr int nsteps = CG::ntaps * NB;
  float bfA[KB][NCJ], afA[KB][NG], bfB[KB][NCJ], afB[KB][NG];
  auto fetch = [&](int s, float (&bf)[KB][NCJ], float (&af)[KB][NG]) {
    const int t = s / NB, cb = (s % NB) * 4 * KB;
    const float* wp = sw + ((size_t)t * KC + cb + lk) * WROW + lr;
    const int ao = CG::tap_off(t) + cb * PS;
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
#pragma unroll
      for (int c = 0; c < NCJ; ++c) bf[kk][c] = wp[4 * kk * WROW + c * 16];
#pragma unroll
      for (int g = 0; g < NG; ++g) af[kk][g] = s_img[abase[g] + ao + 4 * kk * PS];
    }
  };
  auto mma = [&](const float (&bf)[KB][NCJ], const float (&af)[KB][NG]) {
#pragma unroll
    for (int kk = 0; kk < KB; ++kk)
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int c = 0; c < NCJ; ++c)               // rows = channels, cols = pixels; consecutive MFMAs hit different accumulators
          acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[kk][c], af[kk][g], acc[g][c], 0, 0, 0);
  };
  fetch(0, bfA, afA);
  int s = 0;
  if constexpr (KBMAX < 8) {
    // three wavefronts per SIMD (168 registers): keep the k loop a loop -- unrolled over all taps the scheduler hoists the operand
    // fetches of every step to the top and spills
#pragma nounroll
    while (true) {
      if (s + 1 < nsteps) fetch(s + 1, bfB, afB);
      mma(bfA, afA);
      if (++s >= nsteps) break;
      if (s + 1 < nsteps) fetch(s + 1, bfA, afA);
      mma(bfB, afB);
      if (++s >= nsteps) break;
    }
  } else {
    while (true) {
      if (s + 1 < nsteps) fetch(s + 1, bfB, afB);
      mma(bfA, afA);
      if (++s >= nsteps) break;
      if (s + 1 < nsteps) fetch(s + 1, bfA, afA);
      mma(bfB, afB);
      if (++s >= nsteps) break;
    }
  }
}

extern __shared__ __attribute__((aligned(16))) float igemm_smem[];

// grid.x <= number of CUs; block NTHR.  TG: pixel tiles per job sharing the weight fragments (needs NCJ == NCS / 16);
// NCJ: channel tiles per job.  PAIR: process the px = 0 / px = 1 classes of a row parity together and store float2
// (stride-2 layers with an even output width: without it every 64-byte line of y is written twice, half each time --
// measured 401 MB of HBM writes for a 205 MB output).
// PC (producer / consumer; NTHR = 768, double-buffered planes): wavefronts 0..7 only multiply and store -- wavefronts 8..11 (one per
// SIMD, raised priority) fetch the next group from HBM and scatter it into the other plane buffer meanwhile; ONE barrier per group.
// In the other forms all wavefronts walk through fetch, scatter, multiply and store in lockstep, and the matrix pipe idles in every
// phase but one (tools/convt_probe.hip: the MFMA phase is 62 % of a decnn.7 forward wavefront's life and pipe-bound inside).
// STATS: the BatchNorm statistics of the OUTPUT are summed while it is stored and finalised by the last workgroup (bn_sink.hpp).
template <class PL, int IPB, int TG, int NCJ, bool PAIR, int NTHR, bool DB = false, bool PC = false, bool STATS = false>
__global__ __launch_bounds__(NTHR, NTHR == 256 ? 2 : 1) void k_conv_igemm(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int B,
                                                      const float* __restrict__ in_bn, BnSink sink) {
  constexpr int KC = PL::KC, NC = PL::NC, NCS = PL::NCS, NCLS = PL::NCLS, SH = PL::SH, OH = PL::OH, HP = PL::HP, PS = PL::PS;
  constexpr int PADL = PL::PADL, WROW = PL::WROW, NWE = PL::NWE;
  static_assert(NCS % 16 == 0 && NC % NCS == 0 && KC % 4 == 0 && NTHR % 256 == 0, "MFMA tiling");
  static_assert(!PC || (NTHR == 768 && DB), "producer / consumer form: 8 + 4 wavefronts on double-buffered planes");
  constexpr int NW = PC ? 8 : NTHR / 64;             // wavefronts that run the jobs
  constexpr int PT = PC ? 256 : NTHR;                // threads that fetch and scatter the source images
  constexpr int KBM = PC ? 4 : 8;                    // k-steps per operand fetch batch (PC: three wavefronts per SIMD, 168 registers)
  constexpr int NCO = NCS / 16;                      // channel tiles per pass
  static_assert(NCO % NCJ == 0, "channel tiles per job");
  constexpr int NJ = NCO / NCJ;                      // job columns per pixel tile
  static_assert(NJ == 1 || TG == 1, "tile groups share all channel tiles");
  static_assert(!STATS || NJ == 1, "output statistics: a lane keeps the same channels in every job of a pass");
  constexpr int IMG = KC * PS;
  static_assert(IMG % 4 == 0, "float4 zero fill");
  constexpr int SRC = KC * SH * SH;                  // floats per source image
  static_assert(SRC % 4 == 0, "float4 source fetch");
  constexpr int NLD = (IPB * SRC / 4 + PT - 1) / PT;       // float4 fetches per (fetching) thread per group
  constexpr int NBUF = DB ? 2 : 1;
  float* s_img = igemm_smem;                         // [NBUF][IPB][KC][PS] zero padded planes
  float* s_w = igemm_smem + NBUF * IPB * IMG;        // [cls][tap][KC][WROW]
  // in_bn ([KC][4] = mean, invstd, gamma, beta per source channel): the source is the raw output of the previous
  // convolution and the BatchNorm + ReLU that follows it is applied while the images are scattered to LDS, so the
  // normalised activation never goes through HBM (the zero padding is written once and stays zero)
  // The table lives in the unused tail of image 0's channel planes when there is room (plane c holds HP * HP floats of its PS;
  // entry c sits 16-byte aligned behind them, where no operand fetch reaches) -- the footprint is then planes + slabs exactly,
  // which is what lets two 256-thread workgroups of the decnn.7 forward share a CU (2 x 80 KB) -- otherwise behind the slabs.
  constexpr int TFO = (HP * HP + 3) / 4 * 4;
  constexpr bool TF_PAD = igemm_tf_in_pad<PL>();
  float* s_sm = s_w + PL::WSLAB + (TF_PAD ? 0 : 4 * KC);       // STATS: this workgroup's [NC][2] sums, filled pass by pass
  float4* s_tf_tail = reinterpret_cast<float4*>(s_w + PL::WSLAB);
  auto tf_slot = [&](int ch) -> float4* { return TF_PAD ? reinterpret_cast<float4*>(s_img + ch * PS + TFO) : s_tf_tail + ch; };
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int ptid = PC ? tid - 512 : tid;             // index among the fetching threads (PC: negative in the consumers, unused there)
  const int ngroups = (B + IPB - 1) / IPB;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  // wavefront w runs on SIMD w & 3: order the wavefronts SIMD-major so that each SIMD owns a contiguous cost range
  const int jw = (wave & 3) * (NW / 4) + (wave >> 2);

  PROBE_T(pt_all);
  for (int e = tid; e < NBUF * IPB * IMG / 4; e += NTHR) reinterpret_cast<float4*>(s_img)[e] = float4{0.f, 0.f, 0.f, 0.f};
  if (in_bn) {
    if (TF_PAD) __syncthreads();                     // the zero fill above covers the slots
    for (int e = tid; e < KC; e += NTHR) *tf_slot(e) = reinterpret_cast<const float4*>(in_bn)[e];
  }
  if (STATS)
    for (int e = tid; e < 2 * NC; e += NTHR) s_sm[e] = 0.f;

  // NC / NCS passes over the output channels.  When the grid divides evenly the passes are spread over the workgroups
  // (each stages ONE slab and walks every image group with its share of the grid); otherwise every workgroup loops.
  constexpr int NPASS = NC / NCS;
  const bool spread = NPASS > 1 && gridDim.x % NPASS == 0;
  const int gstride = spread ? gridDim.x / NPASS : gridDim.x;       // workgroups walking the image groups of one pass
  const int gfirst = spread ? blockIdx.x % gstride : blockIdx.x;
  for (int pass = spread ? blockIdx.x / gstride : 0; pass < (spread ? blockIdx.x / gstride + 1 : NPASS); ++pass) {
    const int n0 = pass * NCS;
    __syncthreads();                                 // previous pass done with the slabs
    PROBE_T(pt_w);
    for (int base = 0; base < NWE; base += NTHR * 8) {     // batches of 8 loads in flight per thread
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + tid + NTHR * u;
        v[u] = e < NWE ? w[PL::w_src(e, n0)] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + tid + NTHR * u;
        if (e < NWE) s_w[PL::w_dst(e)] = v[u];
      }
    }
    PROBE_ADD(0, pt_w);

    // STATS: shift and running sums of (y - k), (y - k)^2 for the channels this lane stores (the same in every job: NJ == 1)
    float kshift[NCJ][4], bs[NCJ][4], bq[NCJ][4];
#pragma unroll
    for (int cc = 0; cc < NCJ; ++cc)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        kshift[cc][r] = STATS ? bn_sink_shift(sink, n0 + cc * 16 + 4 * lk + r) : 0.f;
        bs[cc][r] = bq[cc][r] = 0.f;
      }
    float4 pre[NLD];
    auto prefetch = [&](int grp) __attribute__((always_inline)) {
      const int b0 = grp * IPB;
      const int nf4 = min(IPB, B - b0) * (SRC / 4);
      const float4* src = x4 + (size_t)b0 * (SRC / 4);
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int f = ptid + PT * i;
        if (PC) pre[i] = src[min(f, nf4 - 1)];       // unconditional (clamped): no branch and no vmcnt(0) per load
        else if (f < nf4) pre[i] = src[f];
      }
    };
    // the prefetched source group -> zero-padded planes (BatchNorm + ReLU on the way when in_bn)
    auto scatter = [&](float* __restrict__ img, int nimg) __attribute__((always_inline)) {
      const int nf4 = nimg * (SRC / 4);
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int f = ptid + PT * i;
        if (f < nf4) {
          const float v[4] = {pre[i].x, pre[i].y, pre[i].z, pre[i].w};
          // one division per float4, then carries: plane / row / column of the following three elements
          int pl = (4 * f) / (SH * SH), q = (4 * f) % (SH * SH), row = q / SH, col = q % SH;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            img[pl * PS + (row + PADL) * HP + col + PADL] = in_bn ? bn_relu(v[k], *tf_slot(pl % KC)) : v[k];
            if (SH % 4 != 0 || k < 3) {              // SH % 4 == 0: the four elements share a row
              if (++col == SH) { col = 0; if (++row == SH) { row = 0; ++pl; } }
            }
          }
        }
      }
    };
    // all (class, pixel tile, channel-tile column) jobs of this wavefront for the group staged in `img`
    // pf_grp >= 0: the fetch of that group is issued after this wavefront's first job (right behind the barrier it would queue
    // behind the previous group's output stores: tools/convt_probe.hip, 4.3k cycles per group)
    auto jobs = [&](const float* __restrict__ img, int b0, int nimg, int pf_grp = -1) {
      PROBE_T(pt_rng);
      // cost (k-steps) of all jobs of the group, and this wavefront's share [lo, hi) of it
      int wtot = 0;
      // class groups: a class on its own, or (PAIR) the two column-parity classes of one row parity, whose pixels
      // interleave along x -- computing both for the same pixel tiles lets a lane store two neighbouring floats
      constexpr int NCG = PAIR ? NCLS / 2 : NCLS;
      static_for<NCG>([&](auto c) {
        constexpr int c0 = PAIR ? 2 * decltype(c)::value : decltype(c)::value;
        using G0 = typename PL::template C<c0>;
        using G1 = typename PL::template C<PAIR ? c0 + 1 : c0>;
        wtot += ((nimg * G0::npc + 15) / 16) * NJ * (G0::ntaps + (PAIR ? G1::ntaps : 0));
      });
      const int lo = (wtot * jw) / NW, hi = (wtot * (jw + 1)) / NW;
      int cbase = 0;
      PROBE_ADD(0, pt_rng);
      static_for<NCG>([&](auto c) {
        constexpr int c0 = PAIR ? 2 * decltype(c)::value : decltype(c)::value;
        using G = typename PL::template C<c0>;                       // px = 0 class of a pair (odd output columns)
        using G1 = typename PL::template C<PAIR ? c0 + 1 : c0>;      // px = 1 class (even output columns)
        static_assert(!PAIR || (G::npc == G1::npc && G::nx == G1::nx), "paired classes cover the same pixel grid");
        static_assert(!PAIR || (G::out_off(0) == G1::out_off(0) + 1 && G::out_off(G::npc - 1) == G1::out_off(G::npc - 1) + 1 &&
                                G1::out_off(0) % 2 == 0 && OH % 2 == 0), "paired classes write neighbouring, 8-byte aligned columns");
        constexpr int ntaps = G::ntaps + (PAIR ? G1::ntaps : 0), npc = G::npc;
        const float* swc = s_w + G::slab_taps * KC * WROW;
        const float* swc1 = s_w + G1::slab_taps * KC * WROW;
        const int mtot = nimg * npc;
        const int njobs = ((mtot + 15) / 16) * NJ;
        // job u of this group belongs to the wavefront whose range holds its cost midpoint
        const int nlo = lo - cbase - ntaps / 2, nhi = hi - cbase - ntaps / 2;
        const int ub = min(njobs, nlo <= 0 ? 0 : (nlo + ntaps - 1) / ntaps), ue = min(njobs, nhi <= 0 ? 0 : (nhi + ntaps - 1) / ntaps);
        cbase += njobs * ntaps;
        for (int u0 = ub; u0 < ue; u0 += TG) {
          PROBE_T(pt_su);
          const int ng = min(TG, ue - u0);
          const int t0 = u0 / NJ, jc = u0 % NJ;      // NJ > 1 implies TG == 1
          // B-operand base address of this lane's pixel in each tile (invalid rows alias pixel 0; masked at the store)
          int abase[TG], abase1[TG];
#pragma unroll
          for (int g = 0; g < TG; ++g) {
            int m = (t0 + g) * 16 + lr;
            m = (g < ng && m < mtot) ? m : 0;
            const int im = m / npc, p = m % npc;
            abase[g] = im * IMG + lk * PS + G::pix_addr(p);
            abase1[g] = im * IMG + lk * PS + G1::pix_addr(p);
          }
          f32x4 acc[TG][NCJ], acc1[TG][NCJ];
#pragma unroll
          for (int g = 0; g < TG; ++g)
#pragma unroll
            for (int cc = 0; cc < NCJ; ++cc) acc[g][cc] = acc1[g][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
          float bv[NCJ][4];                         // in flight during the MFMAs
#pragma unroll
          for (int cc = 0; cc < NCJ; ++cc)
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[cc][r] = bias ? bias[n0 + (jc * NCJ + cc) * 16 + 4 * lk + r] : 0.f;
          PROBE_ADD(6, pt_su);
          PROBE_T(pt_mm);
          auto run = [&](auto gtag, const float* sw, const int (&ab)[TG], f32x4 (&ac)[TG][NCJ]) {
            using GG = typename decltype(gtag)::type;
            switch (ng) {                            // wave-uniform
              case 1: igemm_tile_mma<GG, 1, TG, NCJ, KC, WROW, PS, KBM>(img, sw, ab, lk, lr, ac); break;
              case 2: if constexpr (TG >= 2) igemm_tile_mma<GG, 2, TG, NCJ, KC, WROW, PS, KBM>(img, sw, ab, lk, lr, ac); break;
              case 3: if constexpr (TG >= 3) igemm_tile_mma<GG, 3, TG, NCJ, KC, WROW, PS, KBM>(img, sw, ab, lk, lr, ac); break;
              default: if constexpr (TG >= 4) igemm_tile_mma<GG, 4, TG, NCJ, KC, WROW, PS, KBM>(img, sw, ab, lk, lr, ac); break;
            }
          };
          run(std::common_type<G>{}, swc + jc * NCJ * 16, abase, acc);
          if (pf_grp >= 0) { prefetch(pf_grp); pf_grp = -1; }
          if constexpr (PAIR) run(std::common_type<G1>{}, swc1 + jc * NCJ * 16, abase1, acc1);
          PROBE_ADD(3, pt_mm);
          PROBE_T(pt_st);
          // lane: pixel (t0+g)*16 + lr, channels (jc*NCJ + cc)*16 + 4 lk + r
#pragma unroll
          for (int g = 0; g < TG; ++g) {
            const int m = (t0 + g) * 16 + lr;
            if (g < ng && m < mtot) {
              const int im = m / npc, p = m % npc;
              const int ch0 = n0 + jc * NCJ * 16 + 4 * lk;
              float* yp = y + ((size_t)(b0 + im) * NC + ch0) * (OH * OH) + (PAIR ? G1::out_off(p) : G::out_off(p));
#pragma unroll
              for (int cc = 0; cc < NCJ; ++cc)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const float v0 = acc[g][cc][r] + bv[cc][r];
                  if constexpr (PAIR) { // even column from the px = 1 class, the odd one next to it from px = 0
                    const float v1 = acc1[g][cc][r] + bv[cc][r];
                    *reinterpret_cast<float2*>(yp + (size_t)(cc * 16 + r) * (OH * OH)) = float2{v1, v0};
                    if (STATS) { const float d = v1 - kshift[cc][r]; bs[cc][r] += d; bq[cc][r] = fmaf(d, d, bq[cc][r]); }
                  } else {
                    yp[(size_t)(cc * 16 + r) * (OH * OH)] = v0;
                  }
                  if (STATS) { const float d = v0 - kshift[cc][r]; bs[cc][r] += d; bq[cc][r] = fmaf(d, d, bq[cc][r]); }
                }
            }
          }
          PROBE_ADD(4, pt_st);
        }
      });
      if (pf_grp >= 0) prefetch(pf_grp);             // a wavefront without a job in this group
    };
    if constexpr (PC) {
      const bool producer = wave >= 8;
      int cur = 0;
      if (producer) {
        __builtin_amdgcn_s_setprio(3);               // the youngest wavefront of its SIMD: let it issue whenever it can
        if (gfirst < ngroups) prefetch(gfirst);
      }
      __syncthreads();                               // zero fill, slabs, table
      if (producer) {
        if (gfirst < ngroups) scatter(s_img, min(IPB, B - gfirst * IPB));
        if (gfirst + gstride < ngroups) prefetch(gfirst + gstride);
      }
      __syncthreads();
      for (int grp = gfirst; grp < ngroups; grp += gstride) {
        const int b0 = grp * IPB, nxt = grp + gstride;
        if (producer) {
          if (nxt < ngroups) {
            scatter(s_img + (cur ^ 1) * (IPB * IMG), min(IPB, B - nxt * IPB));
            if (nxt + gstride < ngroups) prefetch(nxt + gstride);
          }
        } else {
          jobs(s_img + cur * (IPB * IMG), b0, min(IPB, B - b0));
        }
        __syncthreads();                             // next buffer complete; this one free for the group after next
        cur ^= 1;
      }
    } else if constexpr (!DB) {
      if (gfirst < ngroups) prefetch(gfirst);
      for (int grp = gfirst; grp < ngroups; grp += gstride) {
        const int b0 = grp * IPB;
        const int nimg = min(IPB, B - b0);
        PROBE_T(pt_b1);
        __syncthreads();                             // previous group's MFMAs have read s_img; zero fill / slabs staged
        PROBE_ADD(1, pt_b1);
        PROBE_T(pt_sc);
        scatter(s_img, nimg);
        __syncthreads();
        PROBE_ADD(2, pt_sc);
        PROBE_T(pt_pf);
        PROBE_ADD(7, pt_pf);
        jobs(s_img, b0, nimg, grp + gstride < ngroups ? grp + gstride : -1);
      }
    } else {
      // Double-buffered planes, ONE barrier per group: while a group is multiplied out of one buffer the next one is
      // scattered into the other.  The two wavefronts of a SIMD (w and w + NW/2) take the scatter at opposite ends of
      // the group -- one before its jobs, one after -- so that on every SIMD one wavefront's scatter / global stores run
      // under the other's MFMAs instead of all eight scattering, multiplying and storing in lockstep.
      const bool early = wave < NW / 2;
      int cur = 0;
      if (gfirst < ngroups) prefetch(gfirst);
      __syncthreads();                               // zero fill, slabs, table
      if (gfirst < ngroups) scatter(s_img, min(IPB, B - gfirst * IPB));
      if (gfirst + gstride < ngroups) prefetch(gfirst + gstride);
      __syncthreads();
      for (int grp = gfirst; grp < ngroups; grp += gstride) {
        const int b0 = grp * IPB, nxt = grp + gstride;
        const int nimg = min(IPB, B - b0);
        float* bc = s_img + cur * (IPB * IMG);
        float* bn = s_img + (cur ^ 1) * (IPB * IMG);
        if (early && nxt < ngroups) {
          scatter(bn, min(IPB, B - nxt * IPB));
          if (nxt + gstride < ngroups) prefetch(nxt + gstride);
        }
        jobs(bc, b0, nimg);
        if (!early && nxt < ngroups) {
          scatter(bn, min(IPB, B - nxt * IPB));
          if (nxt + gstride < ngroups) prefetch(nxt + gstride);
        }
        __syncthreads();                             // next buffer complete; this one free for the group after next
        cur ^= 1;
      }
    }
    if constexpr (STATS) {
      // the pass's sums: over the 16 lanes of a row (one pixel each), then over the wavefronts through the (now idle) slab region
      __syncthreads();
      float* red = s_w;                              // [NW][NCS][2]
#pragma unroll
      for (int cc = 0; cc < NCJ; ++cc)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row_allreduce16(bs[cc][r]), b = row_allreduce16(bq[cc][r]);
          if (lr == 0 && wave < NW) {
            red[(wave * NCS + cc * 16 + 4 * lk + r) * 2] = a;
            red[(wave * NCS + cc * 16 + 4 * lk + r) * 2 + 1] = b;
          }
        }
      __syncthreads();
      if (tid < 2 * NCS) {
        float t = 0.f;
        for (int wv = 0; wv < NW; ++wv) t += red[wv * NCS * 2 + tid];
        s_sm[2 * n0 + tid] = t;
      }
    }
  }
  if constexpr (STATS) {
    __syncthreads();
    bn_sink_publish<NC, NTHR>(sink, s_sm);
  }
  PROBE_ADD(5, pt_all);
}

// ---------------------------------------------------------------------------------------------
// d/d weight of the transposed convolution on the matrix cores:
//   gw[ci][co][ky][kx] = sum_b sum_{iy,ix} x[b][ci][iy][ix] * gy[b][co][S iy - P + ky][S ix - P + kx]
// one GEMM per tap with shared A:  D_tap[ci][co] = sum_k X[ci][k] * G_tap[k][co],  k = (image, pixel), 4 pixels per MFMA.
//   A: lane (lr, lk) supplies x[ci = 16 mt + lr][pixel 4 ks + lk]        (plane stride == 2 mod 4: conflict-free)
//   B: lane (lr, lk) supplies gy[co = 16 nt + lr][window of pixel 4 ks + lk shifted by the tap]; for S = 2 the
//      gy planes are stored split by column parity, so a fixed tap walks the pixels with unit stride.
// PIPE: operands of k-step s+1 fetched into a second register set while the MFMAs of step s issue.
// The 8 wavefronts of a workgroup split the accumulator tiles (WM x WN x WT over ci tiles, co tiles and taps);
// every wavefront walks all k of the workgroup's images.  Persistent grid; the partial sums of a workgroup go to
// part[blockIdx.x][ci][co][tap] and are reduced in a fixed order by k_sum_splits4 (deterministic).
// ---------------------------------------------------------------------------------------------
constexpr int pad2mod4(int n) { return n + ((2 - n % 4) + 4) % 4; }

template <class L> struct WgradGeo {
  static constexpr int NPIX = L::HI * L::HI, NKS = (NPIX + 3) / 4;
  static constexpr int PSX = pad2mod4(NKS * 4);                         // x plane stride (zero tail up to 4 NKS)
  static constexpr int GP = L::GP_;                                     // gy rows / cols, index = oy + P
  static constexpr int GPH = L::S == 2 ? (GP + 1) / 2 : GP;             // columns per (half) plane
  static constexpr int HPL = GP * GPH;                                  // floats per half plane
  static constexpr int PSG = pad2mod4(L::S == 2 ? 2 * HPL : HPL);       // gy plane stride
  static constexpr int IMGX = L::CI * PSX, IMGG = L::CO * PSG;
  static_assert(L::S == 1 || L::S == 2, "stride");
};

template <class L, int IPB, int WM, int WN, int WT, bool PIPE, bool HAS_BN, int NTHR>
__global__ __launch_bounds__(NTHR, NTHR == 256 ? 2 : 1) void k_convT_wgrad_mfma(const float* __restrict__ x, const float* __restrict__ gy,
                                                            float* __restrict__ part, int B, const float* __restrict__ in_bn) {
  using G = WgradGeo<L>;
  constexpr int CI = L::CI, CO = L::CO, HI = L::HI, HO = L::HO, K = L::K, S = L::S, P = L::P, KK = K * K;
  constexpr int MT = CI / 16, NT = CO / 16;
  static_assert(WM * WN * WT * 64 == NTHR && MT % WM == 0 && NT % WN == 0, "wavefront split");
  constexpr int MTW = MT / WM, NTW = NT / WN, NTAPW = (KK + WT - 1) / WT;
  constexpr int NPIX = G::NPIX, NKS = G::NKS, PSX = G::PSX, PSG = G::PSG, GPH = G::GPH, HPL = G::HPL, IMGX = G::IMGX, IMGG = G::IMGG;
  constexpr int SRCX = CI * NPIX, SRCG = CO * HO * HO;
  static_assert(SRCX % 4 == 0 && SRCG % 4 == 0 && (IPB * IMGX) % 4 == 0 && (IPB * IMGG) % 4 == 0, "float4 access");
  constexpr int NLX = (IPB * SRCX / 4 + NTHR - 1) / NTHR, NLG = (IPB * SRCG / 4 + NTHR - 1) / NTHR;
  float* s_x = igemm_smem;                           // [IPB][CI][PSX]
  float* s_g = igemm_smem + IPB * IMGX;              // [IPB][CO][PSG]
  float4* s_tf = reinterpret_cast<float4*>(igemm_smem + IPB * (IMGX + IMGG));   // optional BatchNorm + ReLU of x, as in k_conv_igemm
  if (HAS_BN)
    for (int e = threadIdx.x; e < CI; e += NTHR) s_tf[e] = reinterpret_cast<const float4*>(in_bn)[e];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int wm = wave % WM, wn = (wave / WM) % WN, wt = wave / (WM * WN);
  const int ngroups = (B + IPB - 1) / IPB;

  PROBE_T(pt_all);
  for (int e = tid; e < IPB * (IMGX + IMGG) / 4; e += NTHR) reinterpret_cast<float4*>(igemm_smem)[e] = float4{0.f, 0.f, 0.f, 0.f};

  // LDS offsets of this wavefront's taps (wave-uniform)
  int toff[NTAPW];
#pragma unroll
  for (int j = 0; j < NTAPW; ++j) {
    const int t = min(wt + WT * j, KK - 1), ky = t / K, kx = t % K;
    toff[j] = S == 2 ? (kx & 1) * HPL + ky * GPH + (kx >> 1) : ky * GPH + kx;
  }
  f32x4 acc[MTW][NTW][NTAPW];
#pragma unroll
  for (int a = 0; a < MTW; ++a)
#pragma unroll
    for (int b = 0; b < NTW; ++b)
#pragma unroll
      for (int j = 0; j < NTAPW; ++j) acc[a][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 prex[NLX], preg[NLG];
  auto prefetch = [&](int grp) {
    const int b0 = grp * IPB, nimg = min(IPB, B - b0);
    const float4* sx = reinterpret_cast<const float4*>(x) + (size_t)b0 * (SRCX / 4);
    const float4* sg = reinterpret_cast<const float4*>(gy) + (size_t)b0 * (SRCG / 4);
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int f = tid + NTHR * i;
      if (f < nimg * (SRCX / 4)) prex[i] = sx[f];
    }
#pragma unroll
    for (int i = 0; i < NLG; ++i) {
      const int f = tid + NTHR * i;
      if (f < nimg * (SRCG / 4)) preg[i] = sg[f];
    }
  };
  if ((int)blockIdx.x < ngroups) prefetch(blockIdx.x);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int nimg = min(IPB, B - grp * IPB);
    PROBE_T(pt_b1);
    __syncthreads();                                 // previous group's MFMAs have read the planes; zero fill done
    PROBE_ADD(1, pt_b1);
    PROBE_T(pt_sc);
#pragma unroll
    for (int i = 0; i < NLX; ++i) {
      const int f = tid + NTHR * i;
      if (f < nimg * (SRCX / 4)) {
        const float v[4] = {prex[i].x, prex[i].y, prex[i].z, prex[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = 4 * f + k, pl = e / NPIX;
          s_x[pl * PSX + e % NPIX] = HAS_BN ? bn_relu(v[k], s_tf[pl % CI]) : v[k];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NLG; ++i) {
      const int f = tid + NTHR * i;
      if (f < nimg * (SRCG / 4)) {
        const float v[4] = {preg[i].x, preg[i].y, preg[i].z, preg[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int e = 4 * f + k, pl = e / (HO * HO), q = e % (HO * HO), row = q / HO + P, col = q % HO + P;
          s_g[pl * PSG + (S == 2 ? (col & 1) * HPL + row * GPH + (col >> 1) : row * GPH + col)] = v[k];
        }
      }
    }
    __syncthreads();
    PROBE_ADD(2, pt_sc);
    if (grp + (int)gridDim.x < ngroups) prefetch(grp + gridDim.x);
    PROBE_T(pt_mm);

    // k loop over (image, 4-pixel step), operands of step s+1 in flight while the MFMAs of step s issue
    const int nsteps = nimg * NKS;
    float afA[MTW], bfA[NTAPW][NTW], afB[MTW], bfB[NTAPW][NTW];
    // (addresses are recomputed from s each step: an incremental (row, column) update was measured 17 % slower -- the
    // loop-carried dependence serialises the address arithmetic in front of the LDS fetches)
    auto fetch = [&](int s, float (&af)[MTW], float (&bf)[NTAPW][NTW]) {
      const int im = s / NKS, ks = s % NKS;
      const int p = 4 * ks + lk, pc = min(p, NPIX - 1), iy = pc / HI, ix = pc % HI;   // tail pixels: x is zero there
      const float* xp = s_x + im * IMGX + (wm * MTW * 16 + lr) * PSX + p;
      const float* gp = s_g + im * IMGG + (wn * NTW * 16 + lr) * PSG + (S == 2 ? S * iy * GPH + ix : iy * GPH + ix);
#pragma unroll
      for (int a = 0; a < MTW; ++a) af[a] = xp[a * 16 * PSX];
#pragma unroll
      for (int j = 0; j < NTAPW; ++j)
#pragma unroll
        for (int b = 0; b < NTW; ++b) bf[j][b] = gp[b * 16 * PSG + toff[j]];
    };
    auto mma = [&](const float (&af)[MTW], const float (&bf)[NTAPW][NTW]) {
#pragma unroll
      for (int j = 0; j < NTAPW; ++j)
        if (WT == 1 || wt + WT * j < KK) {           // wave-uniform
#pragma unroll
          for (int a = 0; a < MTW; ++a)
#pragma unroll
            for (int b = 0; b < NTW; ++b) acc[a][b][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[j][b], acc[a][b][j], 0, 0, 0);
        }
    };
    if constexpr (PIPE) {
      fetch(0, afA, bfA);
      int s = 0;
      while (true) {
        if (s + 1 < nsteps) fetch(s + 1, afB, bfB);
        mma(afA, bfA);
        if (++s >= nsteps) break;
        if (s + 1 < nsteps) fetch(s + 1, afA, bfA);
        mma(afB, bfB);
        if (++s >= nsteps) break;
      }
    } else {                                         // register-bound splits: the other wavefront of the SIMD covers the fetch
#pragma unroll 1
      for (int s = 0; s < nsteps; ++s) {
        fetch(s, afA, bfA);
        mma(afA, bfA);
      }
    }
    PROBE_ADD(3, pt_mm);
  }
  // D[m = ci][n = co]: lane holds co = lr, ci = 4 lk + r of each tile.  The partial sums leave in the accumulator
  // layout part[blockIdx.x][tap][mt][nt][r][lane] (64 consecutive floats per store); k_sum_splits_wgrad undoes it.
  float* pp = part + (size_t)blockIdx.x * CI * CO * KK;
#pragma unroll
  for (int j = 0; j < NTAPW; ++j) {
    const int t = wt + WT * j;
    if (t < KK) {
#pragma unroll
      for (int a = 0; a < MTW; ++a)
#pragma unroll
        for (int b = 0; b < NTW; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pp[((((size_t)t * MT + wm * MTW + a) * NT + wn * NTW + b) * 4 + r) * 64 + lane] = acc[a][b][j][r];
    }
  }
  PROBE_ADD(5, pt_all);
}

template <class L, int IPB> constexpr size_t wgrad_lds_bytes() {
  return sizeof(float) * ((size_t)IPB * (WgradGeo<L>::IMGX + WgradGeo<L>::IMGG) + (size_t)4 * L::CI);
}

// gw[ci][co][tap] = sum_s part[s][tap][mt][nt][r][lane] (the accumulator layout of k_convT_wgrad_mfma), fixed order.
// grid = ceil(n / 64), block = 1024 (64 elements x 16 split groups).
static __global__ __launch_bounds__(1024) void k_sum_splits_wgrad(const float* __restrict__ part, int nsplit, int MT, int NT, int KK,
                                                            float* __restrict__ gw) {
  __shared__ float red[16][64];
  const int ex = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const size_t n = (size_t)KK * MT * NT * 256;
  const size_t e = (size_t)blockIdx.x * 64 + ex;
  float acc = 0.f;
  if (e < n) {
#pragma unroll 8
    for (int s = sg; s < nsplit; s += 16) acc += part[(size_t)s * n + e];
  }
  red[sg][ex] = acc;
  __syncthreads();
  if (sg == 0 && e < n) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += red[g][ex];
    const int lane = (int)(e & 63), r = (int)(e >> 6) & 3;
    const size_t tile = e >> 8;                      // (tap * MT + mt) * NT + nt
    const int nt = (int)(tile % NT), mt = (int)((tile / NT) % MT), t = (int)(tile / ((size_t)NT * MT));
    const int ci = mt * 16 + 4 * (lane >> 4) + r, co = nt * 16 + (lane & 15);
    gw[((size_t)ci * (NT * 16) + co) * KK + t] = v;
  }
}

// out[e] = sum_s part[s][e] in a fixed order: 16 interleaved partial sums per element, combined through LDS.
// grid = ceil(n / 64), block = 1024 (64 elements x 16 split groups).
static __global__ __launch_bounds__(1024) void k_sum_splits4(const float* __restrict__ part, int nsplit, size_t n, float* __restrict__ out) {
  __shared__ float red[16][64];
  const int ex = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const size_t e = (size_t)blockIdx.x * 64 + ex;
  float acc = 0.f;
  if (e < n) {
#pragma unroll 8
    for (int s = sg; s < nsplit; s += 16) acc += part[(size_t)s * n + e];
  }
  red[sg][ex] = acc;
  __syncthreads();
  if (sg == 0 && e < n) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += red[g][ex];
    out[e] = v;
  }
}

template <class PL, int IPB, bool DB = false, bool STATS = false> constexpr size_t igemm_lds_bytes() {   // planes + weight slabs (+ the input-transform table when it does not fit the plane tails) (+ the output-statistics sums)
  return sizeof(float) * ((size_t)(DB ? 2 : 1) * IPB * PL::KC * PL::PS + (size_t)PL::WSLAB + (igemm_tf_in_pad<PL>() ? 0 : (size_t)4 * PL::KC) +
                          (STATS ? (size_t)2 * PL::NC : 0));
}

}  // namespace gp

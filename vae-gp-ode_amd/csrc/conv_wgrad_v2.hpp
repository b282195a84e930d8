// conv_wgrad_v2.hpp -- d/d weight of the decoder's transposed convolutions (vae.py:113-121, decnn.4 / decnn.7) on the fp32
// matrix cores, second engine: PRODUCER / CONSUMER wavefronts.
//
//   gw[ci][co][ky][kx] = sum_b sum_{iy,ix} x[b][ci][iy][ix] * gy[b][co][S iy - P + ky][S ix - P + kx]
//
// Same GEMM view, operand layouts and partial-sum layout as k_convT_wgrad_mfma (conv_mfma.hpp): one GEMM per tap with a shared A
// operand, D_tap[ci][co] = sum_k X[ci][k] G_tap[k][co], k = (image, pixel), 4 pixels per v_mfma_f32_16x16x4_f32.  What changed is
// who does what.  In the first engine all eight wavefronts of a workgroup alternate between scattering the next image group into
// LDS (index arithmetic on 70 floats per thread, BatchNorm + ReLU on the way) and multiplying it, in lockstep -- the phase probe
// (profiles/r02a_probes.txt, decnn.7 at 4096 images) shows a wavefront in barriers for 72k and scattering for 27k of its 588k
// cycles, with the matrix pipe idle meanwhile, and the 25 taps dealt 4,3,3,3,3,3,3,3 to the wavefronts (one SIMD gets 14 MFMAs per
// k-step, the others 12).  Here a workgroup is TWELVE wavefronts, three per SIMD:
//   wavefronts 0..7   consumers: nothing but LDS operand fetches and MFMAs.  The (tap, ci tile, co tile) accumulator tiles of the
//                     layer are dealt out as ONE list, contiguous ranges per wavefront with the remainder spread one per SIMD
//                     (decnn.7: 50 tiles -> 7,7,6,6,6,6,6,6, i.e. 13,13,12,12 per SIMD; decnn.4: 200 -> 25 each), so a
//                     wavefront fetches only the A / B fragments its range needs;
//   wavefronts 8..11  producers (one per SIMD): stream image i + 1 from global memory into the OTHER plane buffer while image i is
//                     multiplied -- vector and LDS-store instructions that issue beside the consumers' MFMAs (separate pipes)
//                     instead of in front of them.
// Two plane buffers of one image each (the footprint of the first engine's two-image group), ONE workgroup barrier per image.
#pragma once
#include "conv_mfma.hpp"
#ifndef WGV2_PRIO
#define WGV2_PRIO 3
#endif

namespace gp {

// LDS floats: two plane buffers, the BatchNorm table, the pixel-window table, optionally the producers' scatter table of gy
template <class L> constexpr size_t wgrad_v2_lds_floats(bool gtab) {
  return (size_t)2 * (WgradGeo<L>::IMGX + WgradGeo<L>::IMGG) + (size_t)4 * L::CI + (size_t)4 * (WgradGeo<L>::NKS + 1) +
         (gtab ? (size_t)L::CO * L::HO * L::HO : 0);
}

template <class L> struct WgV2 {
  using G = WgradGeo<L>;
  static constexpr int MT = L::CI / 16, NT = L::CO / 16, KK = L::K * L::K, NUA = KK * MT * NT, NCW = 8;
  // unit u = (tap * MT + mt) * NT + nt; consumer wavefront w owns [ustart(w), ustart(w + 1))
  static constexpr int ustart(int w) {
    int s = 0;
    for (int i = 0; i < w; ++i) s += NUA / NCW + (i < NUA % NCW ? 1 : 0);
    return s;
  }
  static constexpr int tap_of(int u) { return u / (MT * NT); }
  static constexpr int mt_of(int u) { return (u / NT) % MT; }
  static constexpr int nt_of(int u) { return u % NT; }
  static constexpr bool uses_a(int ub, int ue, int mt) {
    for (int u = ub; u < ue; ++u) if (mt_of(u) == mt) return true;
    return false;
  }
  static constexpr bool uses_b(int ub, int ue, int tap, int nt) {
    for (int u = ub; u < ue; ++u) if (tap_of(u) == tap && nt_of(u) == nt) return true;
    return false;
  }
  static constexpr int toff(int t) {                 // LDS offset of tap t inside a gy plane (column-parity split for stride 2)
    return L::S == 2 ? ((t % L::K) & 1) * G::HPL + (t / L::K) * G::GPH + ((t % L::K) >> 1) : (t / L::K) * G::GPH + t % L::K;
  }
};

// all k-steps of one image for the accumulator tiles [UB, UE): operands of step s + 1 in flight while the MFMAs of step s issue.
// s_tab[p] = offset of pixel p's window inside a gy plane (filled once per workgroup): computed per step it costs ~12 vector
// instructions (a clamp, a division by HI, the row / column arithmetic) in front of 6-7 MFMAs, and with two consumers per SIMD
// the SIMD's vector-issue slots, not its matrix pipe, then set the pace (measured: 550 cycles per k-step where the 13 MFMAs of
// the SIMD take 416).  Read from the table two steps ahead, a step is one add, the operand reads and the MFMAs.
template <class L, int UB, int UE, bool PIPE>
__device__ __forceinline__ void wgrad_v2_image(const float* __restrict__ s_x, const float* __restrict__ s_g, const int* __restrict__ s_tab,
                                               int lr, int lk, f32x4 (&acc)[UE - UB]) {
  using W = WgV2<L>;
  using G = typename W::G;
  constexpr int MT = W::MT, NT = W::NT, T0 = W::tap_of(UB), T1 = W::tap_of(UE - 1), NTP = T1 - T0 + 1;
  constexpr int NKS = G::NKS, PSX = G::PSX, PSG = G::PSG;
  float afA[MT], bfA[NTP][NT], afB[MT], bfB[NTP][NT];
  const float* xq = s_x + lr * PSX + lk;             // + 4 ks
  const float* gq = s_g + lr * PSG;                  // + s_tab[4 ks + lk]
  const int* tq = s_tab + lk;
  auto fetch = [&](const float* __restrict__ xp, const float* __restrict__ gp, float (&af)[MT], float (&bf)[NTP][NT]) {
    static_for<MT>([&](auto m) {
      constexpr int mt = decltype(m)::value;
#ifdef WGV2_NOLDS
      if constexpr (W::uses_a(UB, UE, mt)) af[mt] = __int_as_float((int)(size_t)xp + mt);
#else
      if constexpr (W::uses_a(UB, UE, mt)) af[mt] = xp[mt * 16 * PSX];
#endif
    });
    static_for<NTP>([&](auto tt) {
      static_for<NT>([&](auto nn) {
        constexpr int tp = decltype(tt)::value, nt = decltype(nn)::value;
#ifdef WGV2_NOLDS
        if constexpr (W::uses_b(UB, UE, T0 + tp, nt)) bf[tp][nt] = __int_as_float((int)(size_t)gp + tp * 2 + nt);
#else
        if constexpr (W::uses_b(UB, UE, T0 + tp, nt)) bf[tp][nt] = gp[nt * 16 * PSG + W::toff(T0 + tp)];
#endif
      });
    });
  };
  auto mma = [&](const float (&af)[MT], const float (&bf)[NTP][NT]) {
    static_for<UE - UB>([&](auto j) {
      constexpr int u = UB + decltype(j)::value;
      acc[u - UB] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[W::mt_of(u)], bf[W::tap_of(u) - T0][W::nt_of(u)], acc[u - UB], 0, 0, 0);
    });
  };
  if constexpr (PIPE) {
    // Two register sets, alternating, every request UNCONDITIONAL and fenced in front of the MFMAs of the previous step (the
    // machine scheduler otherwise sinks the reads to their first use; with `if (s + 1 < NKS) fetch(...)` the compiler loses count
    // of the reads in flight at the join and waits for lgkmcnt(0) in front of the MFMAs).  No register copies: on this chip the
    // fp32 MFMA runs on the SIMD's vector issue -- the phase probes fit  cycles = 32 x MFMAs + 4 x (every other vector / LDS
    // instruction of ALL wavefronts of the SIMD)  -- so a v_mov beside an MFMA is not free, it is 1/8 of one.
    int g1 = tq[4];
    fetch(xq, gq + tq[0], afA, bfA);
    int s = 0;
#pragma nounroll
    for (; s + 2 < NKS; s += 2) {                    // A holds step s
      fetch(xq + 4 * (s + 1), gq + g1, afB, bfB);
      const int g2 = tq[4 * (s + 2)];
      __builtin_amdgcn_sched_barrier(0);
      mma(afA, bfA);
      __builtin_amdgcn_sched_barrier(0);
      fetch(xq + 4 * (s + 2), gq + g2, afA, bfA);
      g1 = tq[4 * (s + 3)];                          // (table row NKS exists: it repeats row NKS - 1)
      __builtin_amdgcn_sched_barrier(0);
      mma(afB, bfB);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NKS % 2 == 0) {                              // steps NKS - 2 (in A) and NKS - 1
      fetch(xq + 4 * (s + 1), gq + g1, afB, bfB);
      __builtin_amdgcn_sched_barrier(0);
      mma(afA, bfA);
      mma(afB, bfB);
    } else {
      mma(afA, bfA);                                 // step NKS - 1
    }
  } else {                                           // register-bound tile ranges (decnn.4: 25 tiles = 100 accumulator registers under the
#pragma nounroll                                     // 168-register cap of three wavefronts per SIMD): the SIMD's other consumer covers the fetch
    for (int s = 0; s < NKS; ++s) {
      fetch(xq + 4 * s, gq + tq[4 * s], afA, bfA);
      mma(afA, bfA);
    }
  }
}

// The consumer role of wavefront WV: all images of the workgroup for its tile range, then its partial sums.  One copy of the image
// loop per wavefront on purpose -- inside ONE loop that switches on the wavefront, the compiler hoists the loop-invariant operand
// addresses of all eight ranges in front of it and spills them.
template <class L, int WV, bool PIPE>
__device__ __forceinline__ void wgrad_v2_consumer(const float* __restrict__ s_buf, const int* __restrict__ s_tab, int nit, int lane,
                                                  float* __restrict__ pp) {
  using W = WgV2<L>;
  using G = typename W::G;
  constexpr int UB = W::ustart(WV), UE = W::ustart(WV + 1), IMG = G::IMGX + G::IMGG;
  const int lr = lane & 15, lk = lane >> 4;
  f32x4 acc[UE - UB];
#pragma unroll
  for (int j = 0; j < UE - UB; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();                                   // image 0 staged
#ifdef WGV2_PROBE
  long long t_mma = 0, t_bar = 0, t0 = clock64();
  const long long c_start = __builtin_amdgcn_s_memtime(), r_start = __builtin_amdgcn_s_memrealtime();
#endif
#pragma nounroll
  for (int it = 0; it < nit; ++it) {
    const float* cur = s_buf + (it & 1) * IMG;
    wgrad_v2_image<L, UB, UE, PIPE>(cur, cur + G::IMGX, s_tab, lr, lk, acc);
#ifdef WGV2_PROBE
    { const long long t = clock64(); t_mma += t - t0; t0 = t; }
#endif
    __syncthreads();                                 // image it consumed, image it + 1 staged
#ifdef WGV2_PROBE
    { const long long t = clock64(); t_bar += t - t0; t0 = t; }
#endif
  }
#ifdef WGV2_PROBE
  if (blockIdx.x == 0 && lane == 0) {
    const long long dc = __builtin_amdgcn_s_memtime() - c_start, dr = __builtin_amdgcn_s_memrealtime() - r_start;
    printf("wgrad_v2 consumer %d: %d images, mma %lld cycles, barrier wait %lld; loop %lld cycles in %lld ticks of 100 MHz = %.0f MHz\n", WV, nit,
           t_mma, t_bar, dc, dr, 100.0 * (double)dc / (double)dr);
  }
#endif
  // D[m = ci][n = co]: lane holds co = lr, ci = 4 lk + r of each tile.  The partial sums leave in the accumulator layout
  // part[blockIdx.x][unit = (tap, mt, nt)][r][lane] (64 consecutive floats per store); k_sum_splits_wgrad undoes it.
#pragma unroll
  for (int j = 0; j < UE - UB; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) pp[((size_t)(UB + j) * 4 + r) * 64 + lane] = acc[j][r];
}

template <class L, bool HAS_BN, bool PIPE>
__global__ __launch_bounds__(768) void k_convT_wgrad_v2(const float* __restrict__ x, const float* __restrict__ gy,
                                                         float* __restrict__ part, int B, const float* __restrict__ in_bn) {
  using W = WgV2<L>;
  using G = typename W::G;
  constexpr int CI = L::CI, CO = L::CO, HO = L::HO, S = L::S, P = L::P, KK = W::KK;
  constexpr int NPIX = G::NPIX, PSX = G::PSX, PSG = G::PSG, GPH = G::GPH, HPL = G::HPL, IMGX = G::IMGX, IMGG = G::IMGG;
  constexpr int IMG = IMGX + IMGG;                   // one image: x planes, then gy planes
  constexpr int SRCX = CI * NPIX, SRCG = CO * HO * HO, NTHR = 768, NLT = 256;   // NLT producer threads
  static_assert(SRCX % 4 == 0 && SRCG % 4 == 0 && IMG % 4 == 0, "float4 access");
  float* s_buf = igemm_smem;                         // [2][IMG]
  float4* s_tf = reinterpret_cast<float4*>(igemm_smem + 2 * IMG);
  int* s_tab = reinterpret_cast<int*>(igemm_smem + 2 * IMG + 4 * CI);   // [4 (NKS + 1)] window offset of pixel p inside a gy plane
  // producers' scatter table of gy (element of the source image -> offset in the planes), where it is needed and fits
  constexpr bool GTAB = !(S == 2 && HO % 4 == 0 && P == 1) && wgrad_v2_lds_floats<L>(true) * sizeof(float) <= 160 * 1024;
  int* s_gtab = s_tab + 4 * (G::NKS + 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= W::NCW;
  const int lt = tid - 64 * W::NCW;                  // producer thread index
  if (HAS_BN)
    for (int e = tid; e < CI; e += NTHR) s_tf[e] = reinterpret_cast<const float4*>(in_bn)[e];
  if (GTAB)
    for (int e = tid; e < SRCG; e += NTHR) {
      const int pl = e / (HO * HO), q = e % (HO * HO), rr = q / HO + P, cc = q % HO + P;
      s_gtab[e] = pl * PSG + (S == 2 ? (cc & 1) * HPL + rr * GPH + (cc >> 1) : rr * GPH + cc);
    }
  for (int p = tid; p < 4 * (G::NKS + 1); p += NTHR) {
    const int pc = min(p, NPIX - 1), iy = pc / L::HI, ix = pc % L::HI;   // tail pixels: x is zero there
    s_tab[p] = S == 2 ? S * iy * GPH + ix : iy * GPH + ix;
  }
  // the padding of the planes (borders of gy, the tail of x up to a multiple of 4 pixels) is written here and never again
  for (int e = tid; e < 2 * IMG / 4; e += NTHR) reinterpret_cast<float4*>(s_buf)[e] = float4{0.f, 0.f, 0.f, 0.f};
  const int nit = ((int)blockIdx.x < B) ? (B - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;   // images of this workgroup

  __syncthreads();                                   // zero fill, table
  // The two roles run their own loops (one workgroup barrier per image in each: every wavefront passes the same number of them),
  // so that the producers' in-flight image and the consumers' accumulators are never live in the same code.
  if (producer) {
    // The producers are the youngest wavefronts of their SIMDs and would get the issue slots the two MFMA-dense consumers leave
    // over (arbitration is by priority, then age): measured 27k cycles to stage an image the consumers multiply in 23k.  Their work
    // is short, so they go first.
    __builtin_amdgcn_s_setprio(WGV2_PRIO);
    // One image = SRCX / 4 float4 of x, then SRCG / 4 of gy, streamed in chunks of CH float4 per thread: the loads of chunk c + 1
    // are in flight while chunk c is scattered (a load round trip is ~600 cycles here; a whole image in flight per thread costs 72
    // registers and spills under the 168-register cap).  Every load is unconditional (clamped index) and only the LDS stores
    // are predicated, so that the chunk is straight-line code: with the bounds test around the loads the compiler emitted one
    // branch and one `s_waitcnt vmcnt(0)` per load.  Nothing is held across the image barrier.
    constexpr int CH = 4, NX4 = SRCX / 4, NG4 = SRCG / 4, NCX = (NX4 + NLT * CH - 1) / (NLT * CH), NCG = (NG4 + NLT * CH - 1) / (NLT * CH);
    auto load = [&](const float4* __restrict__ src, int n4, int c, float4 (&v)[CH]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < CH; ++i) v[i] = src[min(lt + NLT * (c * CH + i), n4 - 1)];
    };
    auto scatter_x = [&](float* __restrict__ s_x, int c, const float4 (&vv)[CH]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int f = lt + NLT * (c * CH + i);
        const float v[4] = {vv[i].x, vv[i].y, vv[i].z, vv[i].w};
        const bool on = f < NX4;
        int pl = (4 * f) / NPIX, q = (4 * f) % NPIX;             // one division per float4, then carries
        const int pl0 = min(pl, CI - 1);
        float4 tf0 = float4{0.f, 0.f, 0.f, 0.f}, tf1 = tf0;       // the BatchNorm row of the float4's plane, and of the next one
        if (HAS_BN) { tf0 = s_tf[pl0]; if (NPIX % 4 != 0) tf1 = s_tf[min(pl0 + 1, CI - 1)]; }
        if constexpr (NPIX % 4 == 0) {                            // the four pixels share a plane: one base, constant offsets
          float* base = s_x + pl * PSX + q;
          if (on) {
#pragma unroll
            for (int k = 0; k < 4; ++k) base[k] = HAS_BN ? bn_relu(v[k], tf0) : v[k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (on) s_x[pl * PSX + q] = HAS_BN ? bn_relu(v[k], pl == pl0 ? tf0 : tf1) : v[k];
            if (k < 3) { if (++q == NPIX) { q = 0; ++pl; } }
          }
        }
      }
    };
    auto scatter_g = [&](float* __restrict__ s_g, int c, const float4 (&vv)[CH]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const int f = lt + NLT * (c * CH + i);
        const float v[4] = {vv[i].x, vv[i].y, vv[i].z, vv[i].w};
        const bool on = f < NG4;
        int pl = (4 * f) / (HO * HO), q = (4 * f) % (HO * HO), row = q / HO, col = q % HO;
        if constexpr (S == 2 && HO % 4 == 0 && P == 1) {
          // the four columns 4 j .. 4 j + 3 of one row sit at cc = 4 j + 1 .. 4 j + 4 of the padded plane: odd, even, odd, even --
          // positions 2 j, 2 j + 1 of the odd half and 2 j + 1, 2 j + 2 of the even half.  One base address, four constant offsets
          // (the producers' vector instructions come out of the consumers' MFMA time, see wgrad_v2_image).
          float* base = s_g + pl * PSG + (row + P) * GPH + (col >> 1);
          if (on) { base[HPL] = v[0]; base[1] = v[1]; base[HPL + 1] = v[2]; base[2] = v[3]; }
        } else if constexpr (GTAB) {
          // odd widths: the offsets of the four elements come from a table (one 16-byte read) instead of ~40 vector instructions
          const int4 o = reinterpret_cast<const int4*>(s_gtab)[min(f, NG4 - 1)];
          if (on) { s_g[o.x] = v[0]; s_g[o.y] = v[1]; s_g[o.z] = v[2]; s_g[o.w] = v[3]; }
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int rr = row + P, cc = col + P;
            if (on) s_g[pl * PSG + (S == 2 ? (cc & 1) * HPL + rr * GPH + (cc >> 1) : rr * GPH + cc)] = v[k];
            if (HO % 4 != 0 || k < 3) { if (++col == HO) { col = 0; if (++row == HO) { row = 0; ++pl; } } }
          }
        }
      }
    };
    auto stage = [&](int img, float* __restrict__ buf) __attribute__((always_inline)) {
      const float4* sx = reinterpret_cast<const float4*>(x) + (size_t)img * NX4;
      const float4* sg = reinterpret_cast<const float4*>(gy) + (size_t)img * NG4;
      float4 va[CH], vb[CH];
      // x: its few chunks statically (the first gy chunk is requested behind the last of them); gy: a rolled loop over chunk pairs
      load(sx, NX4, 0, va);
      static_for<NCX>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        float4 (&cur)[CH] = (c % 2 == 0) ? va : vb;
        float4 (&nxt)[CH] = (c % 2 == 0) ? vb : va;
        if constexpr (c + 1 < NCX) load(sx, NX4, c + 1, nxt);
        else load(sg, NG4, 0, nxt);
        scatter_x(buf, c, cur);
      });
      float4 (&g0)[CH] = (NCX % 2 == 0) ? va : vb;   // holds gy chunk 0 now
      float4 (&g1)[CH] = (NCX % 2 == 0) ? vb : va;
#pragma nounroll
      for (int c = 0; c < NCG; c += 2) {
        if (c + 1 < NCG) load(sg, NG4, c + 1, g1);
        scatter_g(buf + IMGX, c, g0);
        if (c + 1 < NCG) {
          if (c + 2 < NCG) load(sg, NG4, c + 2, g0);
          scatter_g(buf + IMGX, c + 1, g1);
        }
      }
    };
    if (nit > 0) stage(blockIdx.x, s_buf);
    __syncthreads();                                 // image 0 staged
#ifdef WGV2_PROBE
    long long t_st = 0, t_bar = 0, t0 = clock64();
#endif
    for (int it = 0; it < nit; ++it) {
#ifdef WGV2_NOPROD
      if (it + 1 < 2)
#endif
      if (it + 1 < nit) stage(blockIdx.x + (it + 1) * gridDim.x, s_buf + ((it + 1) & 1) * IMG);
#ifdef WGV2_PROBE
      { const long long t = clock64(); t_st += t - t0; t0 = t; }
#endif
      __syncthreads();                               // image it consumed, image it + 1 staged
#ifdef WGV2_PROBE
      { const long long t = clock64(); t_bar += t - t0; t0 = t; }
#endif
    }
#ifdef WGV2_PROBE
    if (blockIdx.x == 0 && lane == 0) printf("wgrad_v2 producer %d: stage %lld cycles, barrier wait %lld\n", wave, t_st, t_bar);
#endif
    return;
  }
  float* pp = part + (size_t)blockIdx.x * CI * CO * KK;
  switch (wave) {                                    // wave-uniform
    case 0: wgrad_v2_consumer<L, 0, PIPE>(s_buf, s_tab, nit, lane, pp); break;
    case 1: wgrad_v2_consumer<L, 1, PIPE>(s_buf, s_tab, nit, lane, pp); break;
    case 2: wgrad_v2_consumer<L, 2, PIPE>(s_buf, s_tab, nit, lane, pp); break;
    case 3: wgrad_v2_consumer<L, 3, PIPE>(s_buf, s_tab, nit, lane, pp); break;
    case 4: wgrad_v2_consumer<L, 4, PIPE>(s_buf, s_tab, nit, lane, pp); break;
    case 5: wgrad_v2_consumer<L, 5, PIPE>(s_buf, s_tab, nit, lane, pp); break;
    case 6: wgrad_v2_consumer<L, 6, PIPE>(s_buf, s_tab, nit, lane, pp); break;
    default: wgrad_v2_consumer<L, 7, PIPE>(s_buf, s_tab, nit, lane, pp); break;
  }
}

template <class L> constexpr size_t wgrad_v2_lds_bytes() {
  constexpr bool gtab = !(L::S == 2 && L::HO % 4 == 0 && L::P == 1) && wgrad_v2_lds_floats<L>(true) * sizeof(float) <= 160 * 1024;
  return sizeof(float) * wgrad_v2_lds_floats<L>(gtab);
}

}  // namespace gp

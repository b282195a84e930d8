// conv_layers.hpp -- geometry of the conv VAE's transposed-convolution layers (vae.py:52-59, 113-121), shared by the conv translation units.
#pragma once

namespace gp {

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ConvTranspose2d(CI -> CO, K, stride S, padding P, output_padding OP), square HI x HI -> HO x HO
template <int CI_, int CO_, int HI_, int HO_, int OP_, int K_ = 5, int S_ = 2, int P_ = 1> struct CTLayer {
  static constexpr int CI = CI_, CO = CO_, HI = HI_, HO = HO_, OP = OP_, K = K_, S = S_, P = P_;
  static_assert(HO == (HI - 1) * S - 2 * P + K + OP, "geometry");
  // gather form: oy = S qy + py - P, taps ky = py + S t, iy = qy - t
  static constexpr int PL = (K - 1) / S;                             // zero rows/cols before (max t)
  static constexpr int QMAX = (HO - 1 + P) / S;                      // largest qy
  static constexpr int PH = cmax(0, QMAX - (HI - 1));                // zero rows/cols after
  static constexpr int HP = HI + PL + PH;                            // padded input extent
  // channel-plane stride of the staged image: == 16 (mod 32) floats, so that the four 16-lane groups of an MFMA
  // operand fetch (4 consecutive channels x 16 consecutive pixels) fall on disjoint LDS banks
  static constexpr int PS = ((HP * HP + 15) / 32) * 32 + 16;
  static constexpr int GP_ = cmax((HI - 1) * S + K, HO + P);         // padded grad_output extent, index = oy + P
};
using Dec1 = CTLayer<32, 64, 4, 6, 0, 3, 1, 0>;
using Dec4 = CTLayer<64, 32, 6, 13, 0>;
using Dec7 = CTLayer<32, 16, 13, 28, 1>;
using Dec10 = CTLayer<16, 1, 28, 28, 0, 5, 1, 2>;
// the encoder's Conv2d layers, described as the transposed convolutions they are the adjoint of (vae.py:52-59: k5 s2 p2):
// cnn.3 8 -> 16 ch, 14 -> 7 and cnn.6 16 -> 32 ch, 7 -> 4.  Only the directions whose channel counts fill MFMA tiles
// (source channels % 4, output channels % 16) take the matrix-core engine; cnn.0 (1 or 5 input channels) stays generic.
using Enc3 = CTLayer<16, 8, 7, 14, 1, 5, 2, 2>;
using Enc6 = CTLayer<32, 16, 4, 7, 0, 5, 2, 2>;

}  // namespace gp

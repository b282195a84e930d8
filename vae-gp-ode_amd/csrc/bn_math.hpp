// bn_math.hpp -- the BatchNorm affine map with pinned rounding.  Every kernel that evaluates y = relu((x - m) is g + b), or
// only its sign (the ReLU mask of the backward pass), uses these two functions, so that forward and backward agree bit for
// bit on which elements the ReLU passed.
#pragma once
#include <hip/hip_runtime.h>

namespace gp {

__device__ __forceinline__ float bn_xhat(float x, float m, float is) { return __fmul_rn(__fsub_rn(x, m), is); }
__device__ __forceinline__ float bn_affine(float x, float m, float is, float g, float b) { return __fmaf_rn(bn_xhat(x, m, is), g, b); }
// input transform of a convolution that consumes a BatchNorm + ReLU output without materialising it: t = {mean, invstd, gamma, beta}
__device__ __forceinline__ float bn_relu(float x, const float4& t) { return fmaxf(bn_affine(x, t.x, t.y, t.z, t.w), 0.f); }

}  // namespace gp

// Device helpers shared by the compositing kernel and its backward pass.
#pragma once
#include "pr_common.h"

namespace pr {

__device__ __forceinline__ unsigned int float_order_bits(float f) {
    const unsigned int b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// alpha = 1 - exp(-relu(raw) * dist)   (object_composer.py:197)
__device__ __forceinline__ float alpha_of(float raw, float dist) {
    const float relu = raw > 0.f ? raw : 0.f;
    return __fsub_rn(1.0f, expf(__fmul_rn(-relu, dist)));
}

// disparity = 1 / clamp(depth / opacity, min=1e-10), NaN-propagating  (object_composer.py:765)
__device__ __forceinline__ float disparity_of(float depth, float opacity) {
    float q = __fdiv_rn(depth, opacity);
    if (q < 1e-10f) q = 1e-10f;
    return __fdiv_rn(1.0f, q);
}

}  // namespace pr

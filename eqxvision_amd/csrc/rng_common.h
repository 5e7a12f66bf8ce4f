// JAX's Threefry-2x32 bit stream on the device (shared by rng.hip and the attention kernel's dropout branch, attn.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

namespace mv {

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

// Threefry-2x32-20 (Random123; jax/_src/prng.py threefry2x32)
__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
    const uint32_t ks[3] = {k0, k1, k0 ^ k1 ^ 0x1BD11BDAu};
    constexpr int R[2][4] = {{13, 15, 26, 6}, {17, 29, 16, 24}};
    x0 += ks[0];
    x1 += ks[1];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x0 += x1;
            x1 = rotl32(x1, R[i & 1][j]) ^ x0;
        }
        x0 += ks[(i + 1) % 3];
        x1 += ks[(i + 2) % 3] + (uint32_t)(i + 1);
    }
}

// word i of the n-word stream of one key: counters 0 .. n-1 (+ one 0 when n is odd) cut in two halves (x0 | x1)
__device__ __forceinline__ uint32_t stream_word(uint32_t k0, uint32_t k1, uint32_t i, uint32_t n) {
    const uint32_t half = (n + 1) >> 1;
    const bool lo = i < half;
    uint32_t x0 = lo ? i : i - half;
    uint32_t c1 = x0 + half;
    uint32_t x1 = c1 < n ? c1 : 0u;                 // the padding counter
    threefry2x32(k0, k1, x0, x1);
    return lo ? x0 : x1;
}

// jax.random.uniform's float in [0, 1) from a stream word: mantissa bits under exponent 0, minus 1
__device__ __forceinline__ float word_uniform01(uint32_t w) { return __uint_as_float((w >> 9) | 0x3F800000u) - 1.0f; }

}  // namespace mv

// Training-mode Dropout (eqx.nn.Dropout with inference = False, as the reference's classifiers / transformer blocks hold it:
// alexnet.py:63-68, vit.py:39-53, mlps.py:43-52): y = where(bernoulli(key, 1 - p, x.shape), x / (1 - p), 0), one key per sample
// (the caller vmaps over the keys).  The mask is JAX's bit stream (jax.random.bernoulli = uniform(key, shape) < q; uniform =
// mantissa bits of Threefry-2x32 words in the counter layout of `_threefry_random_bits`, see eqxvision_amd/random.py), generated
// here per element: element i of the LOGICAL single-sample array (row-major over the reference's (C,H,W) / (N,D) / (D,) shape)
// is word i of the sample's stream.  HBM-bound (read x, write y); ~110 integer ops per element ride along.
#include "mfma_common.h"

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

template <typename T>
__global__ void dropout_kernel(const T* x, const uint32_t* keys, T* y, long long per, int C, long long HW, int chw, float q,
                               long long total) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const long long b = g / per, pi = g - b * per;      // physical index inside the sample: pixel-major, channel fastest
        long long li = pi;
        if (chw) {                                          // logical (C, H, W) order
            const long long hw = pi / C;
            const int c = (int)(pi - hw * C);
            li = (long long)c * HW + hw;
        }
        const uint32_t w = stream_word(keys[2 * b], keys[2 * b + 1], (uint32_t)li, (uint32_t)per);
        const float u = __uint_as_float((w >> 9) | 0x3F800000u) - 1.0f;
        float v;
        if constexpr (sizeof(T) == 2) v = __uint_as_float((uint32_t)x[g] << 16);
        else v = x[g];
        v = u < q ? v / q : 0.f;
        if constexpr (sizeof(T) == 2) y[g] = (T)(pack_bf2(v, 0.f) & 0xffffu);
        else y[g] = v;
    }
}

}  // namespace mv

using namespace mv;

extern "C" {

int mv_dropout_fwd(const void* x, const void* keys, void* y, int B, int64_t per_sample, int C, int chw_logical, float keep_prob,
                   int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && keys && y, "dropout: NULL pointer");
    MV_CHECK_ARG(B > 0 && per_sample > 0 && per_sample < (1LL << 32) && C > 0 && per_sample % C == 0, "dropout: bad dims B=%d per_sample=%lld C=%d",
                 B, (long long)per_sample, C);
    MV_CHECK_ARG(keep_prob > 0.f && keep_prob <= 1.f, "dropout: keep probability %g outside (0, 1]", keep_prob);
    MV_CHECK_ARG(dtype == MV_BF16 || dtype == MV_F32, "dropout: unknown dtype %d", dtype);
    const long long total = (long long)B * per_sample;
    long long gl = (total + 255) / 256;
    const int grid = (int)(gl > 256 * 16 ? 256 * 16 : gl);
    set_kernel_name("dropout_threefry");
    if (dtype == MV_F32)
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const uint32_t*)keys,
                           (float*)y, (long long)per_sample, C, (long long)(per_sample / C), chw_logical, keep_prob, total);
    else
        hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const uint32_t*)keys,
                           (bf16_t*)y, (long long)per_sample, C, (long long)(per_sample / C), chw_logical, keep_prob, total);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

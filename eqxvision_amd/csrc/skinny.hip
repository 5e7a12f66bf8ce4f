// Skinny Linear for the classifier heads (ResNet fc 2048 -> 1000, ViT / Swin heads 768 -> 1000, AlexNet classifier at
// small batch): M <= 256 rows, gfx950.
//
// The tiled kernels give such a layer one row tile x a handful of channel tiles = 8 blocks on a 256-CU chip (25 us
// for 4 MB of weights).  Here every 32 x 32 output tile is a block of four waves that split the reduction, spread
// over the whole chip (partial sums meet in LDS):
// both MFMA fragments (16 bytes = 8 reduction elements of one weight row / one input row) come straight from global
// memory, eight k16-steps in flight; no LDS, no barrier; fp32 or bf16 output written from the accumulator layout
// (a lane holds 4 consecutive channels of one row: 16-byte stores for fp32).
#include "mfma_common.h"

namespace mv {

struct SkinnyP {
    const bf16_t* x;      // [M][K]
    const bf16_t* w;      // [N][K]
    const float* scale;
    const float* shift;
    void* y;              // [M][N]
    int M, N, K, act, tiles_n;
};

template <typename OutT>
__global__ __launch_bounds__(256) void skinny_linear_kernel(const SkinnyP p) {
    constexpr int D = 8;                                   // k16-steps in flight per wave
    __shared__ float red[3][16][64];                       // partial accumulators of waves 1..3
    const int lane = threadIdx.x & 63, fr = lane & 31, fh = lane >> 5;
    // wave-uniform on purpose: the k-range bounds below must live in SGPRs.  MFMA ignores EXEC, so a bound the
    // compiler believes divergent would turn the tail predicate into a no-op and run the skipped steps anyway.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tn = blockIdx.x % p.tiles_n, tm = blockIdx.x / p.tiles_n;
    const int n0 = tn * 32, m0 = tm * 32;
    const int nr = n0 + fr < p.N ? n0 + fr : p.N - 1;      // clamped rows: never stored
    const int mr = m0 + fr < p.M ? m0 + fr : p.M - 1;
    const bf16_t* wa = p.w + (long long)nr * p.K + fh * 8;
    const bf16_t* xa = p.x + (long long)mr * p.K + fh * 8;
    // the four waves of the block split the reduction: wave w takes k16-steps [kb, ke)
    const int nk = p.K >> 4;
    const int per = (nk + 3) >> 2;
    const int kb = wave * per, ke = (kb + per) < nk ? (kb + per) : nk;
    uint4 af[D], bf[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const int kk = kb + i < nk ? kb + i : nk - 1;
        af[i] = *(const uint4*)(wa + kk * 16);
        bf[i] = *(const uint4*)(xa + kk * 16);
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = kb; k0 < ke; k0 += D) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const uint4 a = af[i], b = bf[i];
            const int kn = k0 + D + i < nk ? k0 + D + i : nk - 1;          // clamped refill (unconditional load)
            af[i] = *(const uint4*)(wa + kn * 16);
            bf[i] = *(const uint4*)(xa + kn * 16);
            if (k0 + i < ke)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                              0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += red[0][e][lane] + red[1][e][lane] + red[2][e][lane];
    // accumulator register e: row (channel) n0 + (e&3) + 8*(e>>2) + 4*fh, column (input row) m0 + fr
    const int m = m0 + fr;
    if (m >= p.M) return;
    OutT* yr = (OutT*)p.y + (long long)m * p.N;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + 8 * g + 4 * fh;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ne = n + e < p.N ? n + e : p.N - 1;
            const float sc = p.scale ? p.scale[ne] : 1.f, sh = p.shift ? p.shift[ne] : 0.f;
            v[e] = fmaf(acc[4 * g + e], sc, sh);
            if (p.act == MV_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
            else if (p.act == MV_ACT_GELU_TANH) v[e] = gelu_tanh_f(v[e]);
        }
        if (n + 3 < p.N && (p.N & 3) == 0) Out4<OutT>::st(yr + n, make_float4(v[0], v[1], v[2], v[3]));
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < p.N) {
                    if constexpr (sizeof(OutT) == 4) ((float*)yr)[n + e] = v[e];
                    else ((bf16_t*)yr)[n + e] = f2bf(v[e]);
                }
        }
    }
}

// fp32 classifier head: x fp32 [M][K], w fp32 [N][K], y fp32 -- `v_mfma_f32_32x32x2_f32` is EXACT fp32 (an fmaf chain) at
// the vector rate, plenty for a 0.2-0.5 GFLOP head.  Why: the logits are the tested quantity; rounding the pooled features and
// the head weights to bf16 alone costs 3-5e-3 of logit error (Swin-T / ViT-B, synthetic weights) out of a 1e-2 budget.
// A lane's 16-byte load holds 4 consecutive k of its row; MFMA step t of the load uses element t from both operands, so the
// k order inside a load is (t, fh) instead of (fh, t) -- the same permutation on both operands, i.e. the same sum.
typedef float f32x16s __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void skinny_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           float* __restrict__ y, int M, int N, int K, int act, int tiles_n) {
    constexpr int D = 4;                                   // 16-byte loads in flight per operand per wave
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63, fr = lane & 31, fh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);       // k-range bounds must be SGPRs (MFMA ignores EXEC)
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int n0 = tn * 32, m0 = tm * 32;
    const int nr = n0 + fr < N ? n0 + fr : N - 1;
    const int mr = m0 + fr < M ? m0 + fr : M - 1;
    const float* wa = w + (long long)nr * K + fh * 4;
    const float* xa = x + (long long)mr * K + fh * 4;
    const int nk = K >> 3;                                 // steps of 8 k (2 lanes halves x 4)
    const int per = (nk + 3) >> 2;
    const int kb = wave * per, ke = (kb + per) < nk ? (kb + per) : nk;
    float4 af[D], bf[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const int kk = kb + i < nk ? kb + i : nk - 1;
        af[i] = *(const float4*)(wa + kk * 8);
        bf[i] = *(const float4*)(xa + kk * 8);
    }
    f32x16s acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = kb; k0 < ke; k0 += D) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const float4 a = af[i], b = bf[i];
            const int kn = k0 + D + i < nk ? k0 + D + i : nk - 1;
            af[i] = *(const float4*)(wa + kn * 8);
            bf[i] = *(const float4*)(xa + kn * 8);
            if (k0 + i < ke) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave - 1][e][lane] = acc[e];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] += red[0][e][lane] + red[1][e][lane] + red[2][e][lane];
    const int m = m0 + fr;
    if (m >= M) return;
    float* yr = y + (long long)m * N;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int n = n0 + 8 * g + 4 * fh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n + e >= N) continue;
            float v = fmaf(acc[4 * g + e], scale ? scale[n + e] : 1.f, shift ? shift[n + e] : 0.f);
            if (act == MV_ACT_RELU) v = fmaxf(v, 0.f);
            else if (act == MV_ACT_GELU_TANH) v = gelu_tanh_f(v);
            yr[n + e] = v;
        }
    }
}

int skinny_f32_supported(long long M, int C, int K, int in_dtype, int out_dtype, const void* residual) {
    return in_dtype == MV_F32 && out_dtype == MV_F32 && residual == nullptr && M >= 1 && M <= 1024 && C % 8 == 0 && C >= 64 &&
           K >= 8 && !get_flag("force_generic");
}

int skinny_f32_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, long long M, int C, int K,
                      int act, hipStream_t st) {
    const int tiles_n = (K + 31) / 32, tiles_m = (int)((M + 31) / 32);
    set_kernel_name("skinny_linear_f32_mfma");
    hipLaunchKernelGGL(skinny_f32_kernel, dim3(tiles_n * tiles_m), dim3(256), 0, st, (const float*)x, (const float*)w, scale, shift,
                       (float*)y, (int)M, K, C, act, tiles_n);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int skinny_supported(long long M, int C, int K, int in_dtype, const void* residual) {
    // C = reduction length, K = output channels (igemm naming)
    return in_dtype == MV_BF16 && residual == nullptr && M >= 1 && M <= 256 && C % 16 == 0 && C >= 64 && K >= 8;
}

int skinny_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, long long M, int C, int K,
                  int act, int out_dtype, hipStream_t st) {
    SkinnyP p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.y = y;
    p.M = (int)M; p.N = K; p.K = C; p.act = act;
    p.tiles_n = (K + 31) / 32;
    const int tiles_m = (int)((M + 31) / 32);
    set_kernel_name("skinny_linear_mfma");
    if (out_dtype == MV_F32)
        hipLaunchKernelGGL(skinny_linear_kernel<float>, dim3(p.tiles_n * tiles_m), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL(skinny_linear_kernel<bf16_t>, dim3(p.tiles_n * tiles_m), dim3(256), 0, st, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // namespace mv

// Squeeze-and-excitation's scale vector in ONE launch (reference layers/squeeze.py:47-60: global mean -> fc1 (1x1 conv) ->
// activation -> fc2 -> scale activation; the multiply x * scale stays the broadcast pass, mv_channel_scale_nhwc_fwd):
//   p[c]  = mean over the HW pixels of x[n, :, c]
//   h[j]  = act1( b1[j] + sum_c w1[j][c] p[c] )          j < S (squeeze channels: 4 ... 320)
//   s[c]  = act2( b2[c] + sum_j w2[c][j] h[j] )          -> scale[n][c] (bf16)        (w2 handed over TRANSPOSED: w2t[S][C])
// One block of 1024 threads per image.  The pooling is what costs (it reads the map once: HBM-bound, as the stand-alone pool);
// the two small matrix-vector products (C x S <= ~1152 x 48 MACs per image) ride in the same block out of LDS, instead of four
// more launches of 6-8 us each (B-row GEMM, a padding copy, an activation pass, a generic-kernel GEMM).  fp32 throughout.
#include "mfma_common.h"

namespace mv {

constexpr int SE_MAX_C = 4096, SE_MAX_S = 512;

__global__ __launch_bounds__(1024) void se_scale_kernel(const uint4* __restrict__ x, const bf16_t* __restrict__ w1, const float* __restrict__ b1,
                                                        const bf16_t* __restrict__ w2, const float* __restrict__ b2, bf16_t* __restrict__ scale,
                                                        int HW, int C, int S, int act1, int act2) {
    extern __shared__ float sm[];                           // red[1024][8] | p[C] | h[S]
    float* red = sm;
    float* pv = sm + 1024 * 8;
    float* hv = pv + C;
    const int n = blockIdx.x, tid = threadIdx.x;
    const int C8 = C >> 3;
    const uint4* xn = x + (long long)n * HW * C8;
    // ---- squeeze: channel vectors in groups of up to 128; the 1024 threads of a group pass = cpb vectors x pl pixel lanes
    for (int cv0 = 0; cv0 < C8; cv0 += 128) {
        const int cpb = (C8 - cv0) < 128 ? (C8 - cv0) : 128;
        const int pl = 1024 / cpb;
        const int cl = tid % cpb, lp = tid / cpb;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (lp < pl) {
            const uint4* xp = xn + cv0 + cl;
            auto acc = [&](const uint4& v) {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[2 * e] += __uint_as_float(w[e] << 16);
                    s[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                }
            };
            int p = lp;
            for (; p + 3 * pl < HW; p += 4 * pl) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = xp[(long long)(p + u * pl) * C8];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc(v[u]);
            }
            for (; p < HW; p += pl) acc(xp[(long long)p * C8]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = s[e];
        __syncthreads();
        for (int i = tid; i < cpb * 8; i += 1024) {         // channel (cv0 + i / 8) * 8 + i % 8: add the pl pixel lanes in order
            const int cv = i >> 3, e = i & 7;
            float t = 0.f;
            for (int q = 0; q < pl; ++q) t += red[(q * cpb + cv) * 8 + e];
            pv[(cv0 + cv) * 8 + e] = t * (1.f / (float)HW);
        }
        __syncthreads();
    }
    // ---- fc1: hidden unit j by wave (j mod 16), lanes over the channels (coalesced rows of w1)
    const int wave = tid >> 6, lane = tid & 63;
    for (int j = wave; j < S; j += 16) {
        const bf16_t* wr = w1 + (long long)j * C;
        float t = 0.f;
        for (int c = lane; c < C; c += 64) t = fmaf(bf2f(wr[c]), pv[c], t);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) hv[j] = apply_act_rt(t + (b1 ? b1[j] : 0.f), act1);
    }
    __syncthreads();
    // ---- fc2 + scale activation: one thread per channel; w2t is [S][C] (transposed by the caller) so that the threads of a wave
    // read consecutive channels of one row
    for (int c = tid; c < C; c += 1024) {
        float t = b2 ? b2[c] : 0.f;
        int j = 0;
        for (; j + 4 <= S; j += 4) {
            const float a0 = bf2f(w2[(long long)j * C + c]), a1 = bf2f(w2[(long long)(j + 1) * C + c]);
            const float a2 = bf2f(w2[(long long)(j + 2) * C + c]), a3 = bf2f(w2[(long long)(j + 3) * C + c]);
            t = fmaf(a0, hv[j], t); t = fmaf(a1, hv[j + 1], t); t = fmaf(a2, hv[j + 2], t); t = fmaf(a3, hv[j + 3], t);
        }
        for (; j < S; ++j) t = fmaf(bf2f(w2[(long long)j * C + c]), hv[j], t);
        scale[(long long)n * C + c] = f2bf(apply_act_rt(t, act2));
    }
}

}  // namespace mv

using namespace mv;

extern "C" {

int mv_se_scale_supported(int C, int S, int dtype) {
    // one block per image does the two matrix-vector products serially after its pooling: past ~64 k MACs per image (MobileNetV3's
    // 672 x 168 and 960 x 240 squeezes) the separate launches -- many blocks each -- win (measured: -7 % on mobilenet_v3_large)
    const long long macs = (long long)C * S;
    return dtype == MV_BF16 && C % 8 == 0 && C >= 8 && C <= SE_MAX_C && S >= 1 && S <= SE_MAX_S && !get_flag("no_se_fused") &&
           (macs <= 65536 || get_flag("se_fused_always"));
}

int mv_se_scale_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* scale, int N, int64_t HW,
                    int C, int S, int act1, int act2, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && w1 && w2 && scale && N > 0 && HW > 0 && HW < (1LL << 31), "se_scale: bad arguments");
    MV_CHECK_ARG(act1 >= MV_ACT_NONE && act1 <= MV_ACT_SILU && act2 >= MV_ACT_NONE && act2 <= MV_ACT_SILU, "se_scale: unknown activation");
    if (!mv_se_scale_supported(C, S, dtype)) {
        set_error("se_scale: unsupported C=%d S=%d dtype=%d (ask mv_se_scale_supported first)", C, S, dtype);
        return MV_E_UNSUPPORTED;
    }
    const size_t smem = (size_t)(1024 * 8 + C + S) * sizeof(float);
    set_kernel_name("se_scale_fused");
    MV_HIP(hipFuncSetAttribute((const void*)se_scale_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(se_scale_kernel, dim3((unsigned)N), dim3(1024), smem, (hipStream_t)stream, (const uint4*)x, (const bf16_t*)w1, b1,
                       (const bf16_t*)w2, b2, (bf16_t*)scale, (int)HW, C, S, act1, act2);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

// Swin patch embedding + its LayerNorm in ONE launch (gfx950):
//     y = LayerNorm2d(Conv2d(3, K, kernel 4, stride 4)(x))            (swin.py:705-711: features[0] = [conv, permute, norm])
// straight from the fp32 NCHW image to the fp32 NHWC residual stream.  The patches do not overlap, so the reduction index
// k = (c, r, s) of a token is three 4 x 4 blocks of the image: a lane's MFMA fragment (8 consecutive k) is two float4 loads (rows
// r, r + 1 of channel c) -- no tap table, no gather.  A wave owns 32 tokens x all K channels (K / 32 accumulator tiles): the
// LayerNorm over K is two register sums + one lane^32 exchange.  The weights are the split-precision pair (hi + lo bf16 terms,
// two MFMAs per step: this layer PRODUCES the residual stream) and live in LDS; the result goes through a
// wave-private LDS patch so that it is stored as whole contiguous rows.  The un-fused pair (generic entry kernel, 8 scalar loads
// per fragment, bf16 result; LayerNorm launch) took 60 + 22 us for 64 images, against 21 us of HBM time.
#include "mfma_common.h"

namespace mv {

namespace {

struct SwinStemP {
    const float* x;        // [B][3][H][W]
    const bf16_t* whi;     // [K][48] (OIHW flattened)
    const bf16_t* wlo;     // low halves, or nullptr
    const float* bias;     // [K] or nullptr
    const float* gamma;    // [K]
    const float* beta;
    float* y;              // [B][H/4][W/4][K]
    int B, H, W;
    long long M;           // tokens
    float eps;
};

template <int K>
__global__ __launch_bounds__(256) void swin_stem_ln_kernel(const SwinStemP p) {
    constexpr int NT = K / 32, WPITCH = 48 * 2 + 16, PPITCH = K * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl[2] = {smem, smem + K * WPITCH};
    char* patch0 = smem + 2 * K * WPITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const bool split = p.wlo != nullptr;
    for (int i = tid; i < K * 6; i += 256) {              // 6 chunks of 16 bytes per weight row
        const int row = i / 6, ch = i - row * 6;
        *(uint4*)(wl[0] + row * WPITCH + ch * 16) = *(const uint4*)(p.whi + row * 48 + ch * 8);
        *(uint4*)(wl[1] + row * WPITCH + ch * 16) = split ? *(const uint4*)(p.wlo + row * 48 + ch * 8) : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const int Ho = p.H / 4, Wo = p.W / 4;
    const long long HW = (long long)p.H * p.W;
    // per-lane epilogue constants: channels 32 t + 8 g + 4 fh .. + 3
    float4 bi[NT][4], ga[NT][4], be[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = 32 * t + 8 * g + 4 * fh;
            bi[t][g] = p.bias ? *(const float4*)(p.bias + ch) : make_float4(0, 0, 0, 0);
            ga[t][g] = *(const float4*)(p.gamma + ch);
            be[t][g] = *(const float4*)(p.beta + ch);
        }
    char* pt = patch0 + wave * (32 * PPITCH);
    const long long ntile = (p.M + 127) / 128;
    for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const long long m0 = tile * 128 + wave * 32;
        long long m = m0 + fr;
        m = m < p.M ? m : p.M - 1;
        const int wo = (int)(m % Wo);
        const long long t_ = m / Wo;
        const int ho = (int)(t_ % Ho), b = (int)(t_ / Ho);
        const float* px = p.x + (long long)b * 3 * HW + (long long)(4 * ho + 2 * fh) * p.W + 4 * wo;
        float4 xv[3][2];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            xv[c][0] = *(const float4*)(px + c * HW);
            xv[c][1] = *(const float4*)(px + c * HW + p.W);
        }
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            uint4 bu;
            bu.x = pack_bf2(xv[c][0].x, xv[c][0].y); bu.y = pack_bf2(xv[c][0].z, xv[c][0].w);
            bu.z = pack_bf2(xv[c][1].x, xv[c][1].y); bu.w = pack_bf2(xv[c][1].z, xv[c][1].w);
            const bf16x8 bfrag = __builtin_bit_cast(bf16x8, bu);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 ah = *(const bf16x8*)(wl[0] + (32 * t + fr) * WPITCH + (2 * c + fh) * 16);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bfrag, acc[t], 0, 0, 0);
                if (split) {
                    const bf16x8 al = *(const bf16x8*)(wl[1] + (32 * t + fr) * WPITCH + (2 * c + fh) * 16);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bfrag, acc[t], 0, 0, 0);
                }
            }
        }
        // bias, LayerNorm over the K channels of my token (my half + the partner lane's), affine
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[t][4 * g + 0] += bi[t][g].x; acc[t][4 * g + 1] += bi[t][g].y;
                acc[t][4 * g + 2] += bi[t][g].z; acc[t][4 * g + 3] += bi[t][g].w;
                s += (acc[t][4 * g + 0] + acc[t][4 * g + 1]) + (acc[t][4 * g + 2] + acc[t][4 * g + 3]);
            }
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / K);
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[t][e] -= mean;
                q += acc[t][e] * acc[t][e];
            }
        q += __shfl_xor(q, 32);
        const float rstd = rsqrtf(q * (1.0f / K) + p.eps);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(pt + fr * PPITCH + (32 * t + 8 * g + 4 * fh) * 4) =
                    make_float4(fmaf(acc[t][4 * g + 0] * rstd, ga[t][g].x, be[t][g].x), fmaf(acc[t][4 * g + 1] * rstd, ga[t][g].y, be[t][g].y),
                                fmaf(acc[t][4 * g + 2] * rstd, ga[t][g].z, be[t][g].z), fmaf(acc[t][4 * g + 3] * rstd, ga[t][g].w, be[t][g].w));
        wave_lds_fence();
        constexpr int QPR = K / 4, NIT = 32 * QPR / 64;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = lane + 64 * i;
            const int r = idx / QPR, qq = idx - r * QPR;
            const float4 v = *(const float4*)(pt + r * PPITCH + qq * 16);
            if (m0 + r < p.M) *(float4*)(p.y + (m0 + r) * K + 4 * qq) = v;
        }
        wave_lds_fence();
    }
}

}  // namespace

}  // namespace mv

extern "C" {

int mv_patch4_ln_supported(int C, int H, int W, int K, int x_dtype) {
    if (mv::get_flag("no_patch4_ln")) return 0;
    return x_dtype == MV_F32 && C == 3 && H % 4 == 0 && W % 4 == 0 && (K == 96 || K == 128);
}

int mv_patch4_ln_fwd(const void* x, const void* w_hi, const void* w_lo, const float* bias, const float* gamma, const float* beta, void* y,
                     int B, int C, int H, int W, int K, float eps, int x_dtype, mv_stream_t stream_) {
    using namespace mv;
    hipStream_t stream = (hipStream_t)stream_;
    MV_CHECK_ARG(x && w_hi && gamma && beta && y && B > 0, "mv_patch4_ln_fwd: null argument");
    if (!mv_patch4_ln_supported(C, H, W, K, x_dtype)) {
        set_error("mv_patch4_ln_fwd: unsupported %dx%dx%d -> %d (ask mv_patch4_ln_supported first)", C, H, W, K);
        return MV_E_UNSUPPORTED;
    }
    SwinStemP p;
    p.x = (const float*)x; p.whi = (const bf16_t*)w_hi; p.wlo = (const bf16_t*)w_lo; p.bias = bias; p.gamma = gamma; p.beta = beta;
    p.y = (float*)y; p.B = B; p.H = H; p.W = W; p.M = (long long)B * (H / 4) * (W / 4); p.eps = eps;
    long long blocks = (p.M + 127) / 128;
    if (blocks > 256 * 4) blocks = 256 * 4;
    set_kernel_name(K == 96 ? "swin_stem_ln_k96" : "swin_stem_ln_k128");
    const int smem = 2 * K * (48 * 2 + 16) + 4 * 32 * (K * 4 + 16);
    if (K == 96) {
        MV_HIP(hipFuncSetAttribute((const void*)swin_stem_ln_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        hipLaunchKernelGGL(swin_stem_ln_kernel<96>, dim3((unsigned)blocks), dim3(256), smem, stream, p);
    } else {
        MV_HIP(hipFuncSetAttribute((const void*)swin_stem_ln_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        hipLaunchKernelGGL(swin_stem_ln_kernel<128>, dim3((unsigned)blocks), dim3(256), smem, stream, p);
    }
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

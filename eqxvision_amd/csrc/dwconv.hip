// Depthwise convolution (groups == channels: `ConvNormActivation(hidden, hidden, groups=hidden)` of the inverted-residual
// blocks, reference mobilenetv2.py:58-68) with the folded BatchNorm and the activation in the same pass, NHWC bf16, gfx950.
//
// There is no reduction over channels, so there is nothing for the matrix cores: 9 multiply-adds per output value against
// 2 + 2 bytes moved -- HBM-bound by an order of magnitude.  One thread owns 8 consecutive channels (16 bytes) of one output
// pixel: per tap one 16-byte load of the input pixel (neighbouring threads = neighbouring channel chunks of the same pixel ->
// whole 128-byte lines; the 3x3 neighbourhood is shared by adjacent output pixels through L1 / L2) and one 16-byte load of the
// tap's weights ([R][S][C] layout: channel-contiguous, a few KB, cache-resident), fp32 accumulation, one 16-byte store.
#include "mfma_common.h"

namespace mv {

struct DwP {
    const bf16_t* x;
    const bf16_t* w;      // [R][S][C]
    const float* scale;   // [C] or null
    const float* shift;   // [C] or null
    bf16_t* y;
    int N, H, W, C, R, S, Ho, Wo, sh, sw, ph, pw, dh, dw, act;
};

__device__ __forceinline__ void unpack8(const uint4 u, float* v) {
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}

__global__ __launch_bounds__(256) void dwconv_kernel(const DwP p) {
    const int C8 = p.C >> 3;
    const long long total = (long long)p.N * p.Ho * p.Wo * C8;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        const int c8 = (int)(r % C8); r /= C8;
        const int wo = (int)(r % p.Wo); r /= p.Wo;
        const int ho = (int)(r % p.Ho);
        const int b = (int)(r / p.Ho);
        const int c = c8 * 8;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int h0 = ho * p.sh - p.ph, w0 = wo * p.sw - p.pw;
        for (int rr = 0; rr < p.R; ++rr) {
            const int hi = h0 + rr * p.dh;
            if ((unsigned)hi >= (unsigned)p.H) continue;
            for (int ss = 0; ss < p.S; ++ss) {
                const int wi = w0 + ss * p.dw;
                if ((unsigned)wi >= (unsigned)p.W) continue;
                float xv[8], wv[8];
                unpack8(*(const uint4*)(p.x + (((long long)b * p.H + hi) * p.W + wi) * p.C + c), xv);
                unpack8(*(const uint4*)(p.w + ((long long)rr * p.S + ss) * p.C + c), wv);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(xv[e], wv[e], acc[e]);
            }
        }
        if (p.scale) {
            const float4 a = *(const float4*)(p.scale + c), b2 = *(const float4*)(p.scale + c + 4);
            acc[0] *= a.x; acc[1] *= a.y; acc[2] *= a.z; acc[3] *= a.w; acc[4] *= b2.x; acc[5] *= b2.y; acc[6] *= b2.z; acc[7] *= b2.w;
        }
        if (p.shift) {
            const float4 a = *(const float4*)(p.shift + c), b2 = *(const float4*)(p.shift + c + 4);
            acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b2.x; acc[5] += b2.y; acc[6] += b2.z; acc[7] += b2.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = apply_act_rt(acc[e], p.act);
        Out8<bf16_t>::st(p.y + idx * 8, acc);
    }
}

// k x k (3 or 5), pad k/2, no dilation (every depthwise layer of MobileNetV2 / V3 / EfficientNet): one thread = 4 consecutive output
// columns x 8 channels.  The KS x ((4-1)*STRIDE + KS) input window is loaded once and every loaded pixel feeds up to KS outputs
// (18 loads for 4 outputs at 3x3 stride 1 instead of 36; 40 instead of 100 at 5x5), a filter row's KS weight vectors once per row.
// WLDS: the KS x KS weight vectors of every channel sit in LDS for the whole block (loaded once) instead of being re-read through
// the vector memory pipe by every thread -- 25 of the 65 loads per thread-iteration at 5x5.
template <int STRIDE, int KS, bool WLDS = false>
__global__ __launch_bounds__(256) void dwconv_kxk_kernel(const DwP p) {
    constexpr int TW = 4, IW = (TW - 1) * STRIDE + KS, PAD = KS / 2;
    extern __shared__ __attribute__((aligned(16))) char wl[];
    if (WLDS) {
        const int nvec = KS * KS * (p.C >> 3);
        for (int i = threadIdx.x; i < nvec; i += 256) ((uint4*)wl)[i] = ((const uint4*)p.w)[i];
        __syncthreads();
    }
    const int C8 = p.C >> 3, WT = (p.Wo + TW - 1) / TW;
    const long long total = (long long)p.N * p.Ho * WT * C8;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        const int c8 = (int)(r % C8); r /= C8;
        const int wt = (int)(r % WT); r /= WT;
        const int ho = (int)(r % p.Ho);
        const int b = (int)(r / p.Ho);
        const int c = c8 * 8, wo0 = wt * TW;
        float acc[TW][8];
#pragma unroll
        for (int o = 0; o < TW; ++o)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = 0.f;
        const int h0 = ho * STRIDE - PAD, w0 = wo0 * STRIDE - PAD;
#pragma unroll
        for (int rr = 0; rr < KS; ++rr) {
            const int hi = h0 + rr;
            if ((unsigned)hi >= (unsigned)p.H) continue;
            float wv[KS][8];
#pragma unroll
            for (int t = 0; t < KS; ++t) {
                if (WLDS) unpack8(*(const uint4*)(wl + ((rr * KS + t) * p.C + c) * 2), wv[t]);
                else unpack8(*(const uint4*)(p.w + (long long)(rr * KS + t) * p.C + c), wv[t]);
            }
            const bf16_t* row = p.x + ((long long)b * p.H + hi) * p.W * p.C + c;
#pragma unroll
            for (int j = 0; j < IW; ++j) {
                const int wi = w0 + j;
                if ((unsigned)wi >= (unsigned)p.W) continue;
                float xv[8];
                unpack8(*(const uint4*)(row + (long long)wi * p.C), xv);
#pragma unroll
                for (int o = 0; o < TW; ++o) {
                    const int ss = j - o * STRIDE;                   // tap column of output o that this input column feeds
                    if (ss >= 0 && ss < KS) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(xv[e], wv[ss][e], acc[o][e]);
                    }
                }
            }
        }
        float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, sf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.scale) {
            const float4 a = *(const float4*)(p.scale + c), b2 = *(const float4*)(p.scale + c + 4);
            sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b2.x; sc[5] = b2.y; sc[6] = b2.z; sc[7] = b2.w;
        }
        if (p.shift) {
            const float4 a = *(const float4*)(p.shift + c), b2 = *(const float4*)(p.shift + c + 4);
            sf[0] = a.x; sf[1] = a.y; sf[2] = a.z; sf[3] = a.w; sf[4] = b2.x; sf[5] = b2.y; sf[6] = b2.z; sf[7] = b2.w;
        }
#pragma unroll
        for (int o = 0; o < TW; ++o) {
            if (wo0 + o >= p.Wo) break;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[o][e] = apply_act_rt(fmaf(acc[o][e], sc[e], sf[e]), p.act);
            Out8<bf16_t>::st(p.y + (((long long)b * p.Ho + ho) * p.Wo + wo0 + o) * p.C + c, acc[o]);
        }
    }
}

// Stride 1, k = 3 / 5, pad k/2: the input window of an 8 x 16 output tile and a slab of 64 channels (8 vectors) goes through LDS --
// every input pixel is fetched from memory ONCE per tile (coalesced 128-byte pieces) instead of up to k x k times through L1 by the
// threads that need it (40 vector loads per 4 outputs at 5x5: the kernel above runs at 1.1-1.3 TB/s there).  256 threads: thread =
// (vector v of the slab, 4 consecutive output columns of one row); pixel pitch 144 bytes in LDS: the 8 pixel groups a wave reads at
// once start 64 bytes apart modulo the 256-byte bank period (conflict-free 16-byte reads).  The filter vectors of the slab sit in
// LDS too.  What bounds it then is the fp32 FMA count (25 per value at 5x5).
template <int KS>
__global__ __launch_bounds__(256) void dwconv_tile_kernel(const DwP p) {
    constexpr int TH = 8, TWB = 16, PAD = KS / 2, IH = TH + KS - 1, IWB = TWB + KS - 1, PP = 144;
    __shared__ __attribute__((aligned(16))) char xl[IH * IWB * PP];
    __shared__ __attribute__((aligned(16))) uint4 wl[KS * KS * 8];
    const int C8 = p.C >> 3;
    const int tiles_w = (p.Wo + TWB - 1) / TWB;
    const int th = blockIdx.x / tiles_w, tw = blockIdx.x - th * tiles_w;
    const int v0 = blockIdx.y * 8, b = blockIdx.z;
    const int nvec = (C8 - v0) < 8 ? (C8 - v0) : 8;
    const int ho0 = th * TH, wo0 = tw * TWB;
    const int tid = threadIdx.x;
    // ---- filter vectors and the input window -> LDS (zeros outside the image / past the last channel vector)
    for (int i = tid; i < KS * KS * 8; i += 256) {
        const int t = i >> 3, v = i & 7;
        wl[i] = v < nvec ? *(const uint4*)(p.w + (long long)t * p.C + (v0 + v) * 8) : make_uint4(0, 0, 0, 0);
    }
    const bf16_t* xb = p.x + (long long)b * p.H * p.W * p.C;
    for (int i = tid; i < IH * IWB * 8; i += 256) {
        const int pix = i >> 3, v = i & 7;
        const int iy = pix / IWB, ix = pix - iy * IWB;
        const int hi = ho0 - PAD + iy, wi = wo0 - PAD + ix;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (v < nvec && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
            u = *(const uint4*)(xb + ((long long)hi * p.W + wi) * p.C + (v0 + v) * 8);
        *(uint4*)(xl + pix * PP + v * 16) = u;
    }
    __syncthreads();
    const int v = tid & 7, q = tid >> 3;
    const int row = q >> 2, col0 = (q & 3) * 4;
    if (v >= nvec) return;
    float acc[4][8];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = 0.f;
#pragma unroll 1
    for (int rr = 0; rr < KS; ++rr) {                       // (not unrolled: the whole window in registers costs the occupancy)
        float wv[KS][8];
#pragma unroll
        for (int t = 0; t < KS; ++t) unpack8(wl[(rr * KS + t) * 8 + v], wv[t]);
        const char* xr = xl + ((row + rr) * IWB + col0) * PP + v * 16;
#pragma unroll
        for (int j = 0; j < 4 + KS - 1; ++j) {
            float xv[8];
            unpack8(*(const uint4*)(xr + j * PP), xv);
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int ss = j - o;
                if (ss >= 0 && ss < KS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[o][e] = fmaf(xv[e], wv[ss][e], acc[o][e]);
                }
            }
        }
    }
    const int c = (v0 + v) * 8, ho = ho0 + row;
    if (ho >= p.Ho) return;
    float sc[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f}, sf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.scale) {
        const float4 a = *(const float4*)(p.scale + c), b2 = *(const float4*)(p.scale + c + 4);
        sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b2.x; sc[5] = b2.y; sc[6] = b2.z; sc[7] = b2.w;
    }
    if (p.shift) {
        const float4 a = *(const float4*)(p.shift + c), b2 = *(const float4*)(p.shift + c + 4);
        sf[0] = a.x; sf[1] = a.y; sf[2] = a.z; sf[3] = a.w; sf[4] = b2.x; sf[5] = b2.y; sf[6] = b2.z; sf[7] = b2.w;
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const int wo = wo0 + col0 + o;
        if (wo >= p.Wo) break;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o][e] = apply_act_rt(fmaf(acc[o][e], sc[e], sf[e]), p.act);
        Out8<bf16_t>::st(p.y + (((long long)b * p.Ho + ho) * p.Wo + wo) * p.C + c, acc[o]);
    }
}

int dwconv_supported(int C, int K, int groups, int R, int S, int in_dtype, int out_dtype) {
    return groups == C && K == C && C % 8 == 0 && R * S <= 49 && in_dtype == MV_BF16 && out_dtype == MV_BF16;
}

int dwconv_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int H, int W, int C, int R,
                  int S, int sh, int sw, int ph, int pw, int dh, int dw, int act, hipStream_t st) {
    DwP p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.y = (bf16_t*)y;
    p.N = N; p.H = H; p.W = W; p.C = C; p.R = R; p.S = S;
    p.Ho = (H + 2 * ph - dh * (R - 1) - 1) / sh + 1;
    p.Wo = (W + 2 * pw - dw * (S - 1) - 1) / sw + 1;
    p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw; p.act = act;
    const bool fast = R == S && (R == 3 || R == 5) && ph == R / 2 && pw == R / 2 && dh == 1 && dw == 1 && sh == sw && (sh == 1 || sh == 2) &&
                      !get_flag("dwconv_generic");
    const long long total = (long long)N * p.Ho * (fast ? (p.Wo + 3) / 4 : p.Wo) * (C / 8);
    long long g = (total + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    // stride 1, 5 x 5: the LDS-tiled kernel.  (3 x 3 too with flag "dwconv_tile3": its window overlaps little enough that L1 serves the
    // register-window kernel well; the tile's 40 % halo and barrier cost more: -2.5 % / -4 % on mobilenet_v2 / v3.)
    if (fast && sh == 1 && N <= 65535 && !get_flag("dwconv_no_tile") && (R == 5 || get_flag("dwconv_tile3"))) {
        char name[48];
        snprintf(name, sizeof(name), "dwconv%dx%d_s1_lds_tile", R, R);
        set_kernel_name(name);
        const dim3 grid((unsigned)(((p.Ho + 7) / 8) * ((p.Wo + 15) / 16)), (unsigned)((C / 8 + 7) / 8), (unsigned)N);
        if (R == 3) hipLaunchKernelGGL((dwconv_tile_kernel<3>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((dwconv_tile_kernel<5>), grid, dim3(256), 0, st, p);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    if (fast) {
        char name[48];
        snprintf(name, sizeof(name), "dwconv%dx%d_s%d_bf16x8x4", R, R, sh);
        set_kernel_name(name);
        const dim3 grid((unsigned)g), block(256);
        const size_t wbytes = (size_t)R * R * C * 2;
        if (wbytes <= 40 * 1024) {        // weights in LDS (<= 40 KB: still four blocks per CU)
            if (R == 3 && sh == 1) hipLaunchKernelGGL((dwconv_kxk_kernel<1, 3, true>), grid, block, wbytes, st, p);
            else if (R == 3) hipLaunchKernelGGL((dwconv_kxk_kernel<2, 3, true>), grid, block, wbytes, st, p);
            else if (sh == 1) hipLaunchKernelGGL((dwconv_kxk_kernel<1, 5, true>), grid, block, wbytes, st, p);
            else hipLaunchKernelGGL((dwconv_kxk_kernel<2, 5, true>), grid, block, wbytes, st, p);
            MV_LAUNCH_CHECK();
            return MV_OK;
        }
        if (R == 3 && sh == 1) hipLaunchKernelGGL((dwconv_kxk_kernel<1, 3>), grid, block, 0, st, p);
        else if (R == 3) hipLaunchKernelGGL((dwconv_kxk_kernel<2, 3>), grid, block, 0, st, p);
        else if (sh == 1) hipLaunchKernelGGL((dwconv_kxk_kernel<1, 5>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((dwconv_kxk_kernel<2, 5>), grid, block, 0, st, p);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    set_kernel_name("dwconv_nhwc_bf16x8");
    hipLaunchKernelGGL(dwconv_kernel, dim3((unsigned)g), dim3(256), 0, st, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // namespace mv

// Shape-agnostic VALU kernels (any size, f32 or bf16) + the memory-bound ops (pooling,
// LayerNorm, layout converters) + the C-ABI entry points that dispatch between these and the
// MFMA kernels in igemm.hip / stem.hip / attn.hip.
//
// The contraction kernels here are the "any shape" path (odd channel counts such as the
// MlpProjection(20,5,10) of the reference's tests/test_layers.py:25-40) and the on-device
// cross-check for the MFMA kernels (flag "force_generic").  They are NOT the fast path.
#include "mfma_common.h"

namespace mv {

// ------------------------------------------------------------------------------------------
// direct convolution, one thread per (output row, output channel); x/w addressed by strides so
// the same kernel serves NHWC activations + KRSC weights and NCHW images + OIHW weights.
// ------------------------------------------------------------------------------------------
struct ConvP {
    int N, H, W, C, K, R, S, Ho, Wo, sh, sw, ph, pw, dh, dw, groups;
    long long sxn, sxc, sxh, sxw;  // x strides (elements)
    long long swk, swc, swr, sws;  // w strides (elements)
    int act, tok_stride, tok_offset;
};

template <typename TX, typename TW, typename TY>
__global__ void conv_generic_kernel(const TX* __restrict__ x, const TW* __restrict__ w,
                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                    const TY* __restrict__ residual, TY* __restrict__ y,
                                    const float* __restrict__ pos, ConvP p) {
    const int k = blockIdx.y * blockDim.x + threadIdx.x;
    const long long m = (long long)blockIdx.x * blockDim.y + threadIdx.y;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    if (k >= p.K || m >= M) return;
    const int wo = (int)(m % p.Wo);
    const int ho = (int)((m / p.Wo) % p.Ho);
    const int n = (int)(m / ((long long)p.Wo * p.Ho));
    const int Cg = p.C / p.groups, Kg = p.K / p.groups;
    const int g = k / Kg;
    float acc = 0.f;
    for (int r = 0; r < p.R; ++r) {
        const int hi = ho * p.sh - p.ph + r * p.dh;
        if (hi < 0 || hi >= p.H) continue;
        for (int s = 0; s < p.S; ++s) {
            const int wi = wo * p.sw - p.pw + s * p.dw;
            if (wi < 0 || wi >= p.W) continue;
            const TX* xp = x + n * p.sxn + hi * p.sxh + wi * p.sxw + (long long)(g * Cg) * p.sxc;
            const TW* wp = w + k * p.swk + r * p.swr + s * p.sws;
            for (int c = 0; c < Cg; ++c) acc = fmaf(io<TX>::ld(xp + c * p.sxc), io<TW>::ld(wp + c * p.swc), acc);
        }
    }
    float v = acc;
    if (scale) v *= scale[k];
    if (shift) v += shift[k];
    long long row = m;
    if (p.tok_stride > 0) {
        const int pix = ho * p.Wo + wo;
        row = (long long)n * p.tok_stride + p.tok_offset + pix;
        if (pos) v += pos[(long long)(p.tok_offset + pix) * p.K + k];
    }
    if (residual) v += io<TY>::ld(residual + row * p.K + k);
    v = apply_act_rt(v, p.act);
    io<TY>::st(y + row * p.K + k, v);
}

template <typename TX, typename TW, typename TY>
static int conv_generic_launch(const void* x, const void* w, const float* scale, const float* shift,
                               const void* residual, void* y, const float* pos, const ConvP& p, hipStream_t st) {
    const long long M = (long long)p.N * p.Ho * p.Wo;
    dim3 block(64, 4);
    dim3 grid((unsigned)((M + 3) / 4), (p.K + 63) / 64);
    hipLaunchKernelGGL((conv_generic_kernel<TX, TW, TY>), grid, block, 0, st, (const TX*)x, (const TW*)w, scale, shift,
                       (const TY*)residual, (TY*)y, pos, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

// ------------------------------------------------------------------------------------------
// the same contraction in fp32 on the matrix cores (v_mfma_f32_32x32x2_f32): the fp32 compute mode and the training step (both
// passes run in fp32) spend their time here.  One wave = 32 output channels x 32 output positions; A = weights (lane: channel fr,
// reduction element fh), B = the positions' input rows; a lane loads FOUR consecutive reduction channels at once (its half of a group
// of eight), which feeds four MFMA steps when the reduction index is contiguous in both operands (channels-last); any other layout
// (the NCHW image + OIHW filter of the entry convolutions) takes two strided scalars per step.  groups = 1.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_f32_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ residual, float* __restrict__ y,
                                                            const float* __restrict__ pos, ConvP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 31, fh = lane >> 5;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const long long m = ((long long)blockIdx.x * 4 + wave) * 32 + fr;
    const int k0 = blockIdx.y * 32;
    const bool mok = m < M;
    const long long mm = mok ? m : 0;
    const int wo = (int)(mm % p.Wo), ho = (int)((mm / p.Wo) % p.Ho), n = (int)(mm / ((long long)p.Wo * p.Ho));
    const bool kok = k0 + fr < p.K;
    const float* wk = w + (long long)(kok ? k0 + fr : 0) * p.swk;
    const bool vec = p.sxc == 1 && p.swc == 1 && (p.C & 3) == 0 && (p.sxn & 3) == 0 && (p.sxh & 3) == 0 && (p.sxw & 3) == 0 &&
                     (p.swk & 3) == 0 && (p.swr & 3) == 0 && (p.sws & 3) == 0;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int r = 0; r < p.R; ++r) {
        const int hi = ho * p.sh - p.ph + r * p.dh;
        for (int s_ = 0; s_ < p.S; ++s_) {
            const int wi = wo * p.sw - p.pw + s_ * p.dw;
            const bool inside = mok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const float* xp = x + (inside ? n * p.sxn + hi * p.sxh + wi * p.sxw : 0);
            const float* wp = wk + r * p.swr + s_ * p.sws;
            if (vec) {
                for (int c0 = 0; c0 < p.C; c0 += 8) {
                    const int c = c0 + 4 * fh;
                    const bool cin = c < p.C;                                    // C % 8 == 4: the upper half of the last group is empty
                    const float4 a = (kok && cin) ? *(const float4*)(wp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 b = (inside && cin) ? *(const float4*)(xp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
                }
            } else {
                for (int c0 = 0; c0 < p.C; c0 += 2) {
                    const int c = c0 + fh;
                    const bool cin = c < p.C;
                    const float a = (kok && cin) ? wp[c * p.swc] : 0.f;       // any layout: NCHW images + OIHW filters (the entry convolutions)
                    const float b = (inside && cin) ? xp[c * p.sxc] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
        }
    }
    if (!mok) return;
    long long row = m;
    int pix = 0;
    if (p.tok_stride > 0) {
        pix = ho * p.Wo + wo;
        row = (long long)n * p.tok_stride + p.tok_offset + pix;
    }
    // acc[4 q + i]: channel k0 + 8 q + 4 fh + i of position fr
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + 8 * q + 4 * fh + i;
            if (k >= p.K) continue;
            float v = acc[4 * q + i];
            if (scale) v *= scale[k];
            if (shift) v += shift[k];
            if (p.tok_stride > 0 && pos) v += pos[(long long)(p.tok_offset + pix) * p.K + k];
            if (residual) v += residual[row * p.K + k];
            v = apply_act_rt(v, p.act);
            y[row * p.K + k] = v;
        }
}

// The same contraction with both operands staged through LDS: a block = 64 channels x 256 positions, eight waves of 64 x 32 (two MFMA
// tiles each); the reduction runs in chunks of 32 channels of one filter tap: 8 KB of weights + 32 KB of input rows per chunk, fetched
// as whole 128-byte rows (eight lanes per row), double-buffered (the next chunk travels global -> registers while this one
// multiplies, then registers -> LDS behind one barrier).  Rows are padded to 36 floats: conflict-free ds_read_b128 fragments.
// Needs 16-byte-aligned rows (C and all strides multiples of 4); dispatched from 64 reduction channels per tap up (with four waves per
// block it only paid from 128; eight waves: resnet50 fp32 4291 -> 5097 img/s, vit_base 1414 -> 1585, swin_t 3214 -> 3697).
__global__ __launch_bounds__(512) void conv_f32_lds_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ residual, float* __restrict__ y,
                                                           const float* __restrict__ pos, ConvP p) {
    constexpr int RP = 36, WROWS = 64, XROWS = 256, STAGE = (WROWS + XROWS) * RP;       // floats per stage
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 31, fh = lane >> 5;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const long long m0 = (long long)blockIdx.x * XROWS;
    const int k0 = blockIdx.y * WROWS;
    // loader role: thread t fetches float4 number (t & 7) of rows (t >> 3) + 64 i (eight waves: two per SIMD)
    const int lrow = tid >> 3, lq = tid & 7;
    int xn[4], xho[4], xwo[4];
    bool xok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long m = m0 + lrow + 64 * i;
        xok[i] = m < M;
        const long long mm = xok[i] ? m : 0;
        xwo[i] = (int)(mm % p.Wo); xho[i] = (int)((mm / p.Wo) % p.Ho); xn[i] = (int)(mm / ((long long)p.Wo * p.Ho));
    }
    const int cpt = (p.C + 31) / 32;                                   // chunks per filter tap
    const int nchunk = p.R * p.S * cpt;
    float4 gx[4], gw;
    auto fetch = [&](int ch) {
        const int tap = ch / cpt, c0 = (ch - tap * cpt) * 32 + 4 * lq;
        const int r = tap / p.S, s_ = tap - r * p.S;
        const bool cin = c0 < p.C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int hi = xho[i] * p.sh - p.ph + r * p.dh, wi = xwo[i] * p.sw - p.pw + s_ * p.dw;
            const bool in = xok[i] && cin && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            gx[i] = in ? *(const float4*)(x + xn[i] * p.sxn + hi * p.sxh + wi * p.sxw + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        {
            const int k = k0 + lrow;
            gw = (k < p.K && cin) ? *(const float4*)(w + (long long)k * p.swk + r * p.swr + s_ * p.sws + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int buf) {
        float* base = lds + buf * STAGE;
        *(float4*)(base + lrow * RP + 4 * lq) = gw;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(float4*)(base + (WROWS + lrow + 64 * i) * RP + 4 * lq) = gx[i];
    };
    f32x16 acc[2];                                                     // a wave: 64 channels x 32 positions
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunk) fetch(ch + 1);
        const float* wl = lds + buf * STAGE + fr * RP + 4 * fh;
        const float* xl = lds + buf * STAGE + (WROWS + 32 * wave + fr) * RP + 4 * fh;
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                   // 8 reduction channels per step: a lane holds its half of 4
            float4 a[2];
            const float4 b = *(const float4*)(xl + 8 * j);
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = *(const float4*)(wl + 32 * t * RP + 8 * j);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kt].x, b.x, acc[kt], 0, 0, 0);
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kt].y, b.y, acc[kt], 0, 0, 0);
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kt].z, b.z, acc[kt], 0, 0, 0);
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kt].w, b.w, acc[kt], 0, 0, 0);
            }
        }
        if (ch + 1 < nchunk) {
            stash(buf ^ 1);                                             // its last reader finished before the previous barrier
            __syncthreads();
        }
    }
    // epilogue: acc[kt][4 q + i] = channel k0 + 32 kt + 8 q + 4 fh + i of position m0 + 32 wave + fr
    {
        const long long m = m0 + 32 * wave + fr;
        if (m >= M) return;
        long long row = m;
        int pix = 0;
        if (p.tok_stride > 0) {
            const int wo = (int)(m % p.Wo), ho = (int)((m / p.Wo) % p.Ho), n = (int)(m / ((long long)p.Wo * p.Ho));
            pix = ho * p.Wo + wo;
            row = (long long)n * p.tok_stride + p.tok_offset + pix;
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = k0 + 32 * kt + 8 * q + 4 * fh + i;
                    if (k >= p.K) continue;
                    float v = acc[kt][4 * q + i];
                    if (scale) v *= scale[k];
                    if (shift) v += shift[k];
                    if (p.tok_stride > 0 && pos) v += pos[(long long)(p.tok_offset + pix) * p.K + k];
                    if (residual) v += residual[row * p.K + k];
                    v = apply_act_rt(v, p.act);
                    y[row * p.K + k] = v;
                }
    }
}

static int conv_generic_dispatch(const void* x, const void* w, const float* scale, const float* shift,
                                 const void* residual, void* y, const float* pos, const ConvP& p, int x_dtype,
                                 int w_dtype, int y_dtype, hipStream_t st) {
    if (x_dtype == MV_F32 && w_dtype == MV_F32 && y_dtype == MV_F32 && p.groups == 1 && p.sxc == 1 && p.swc == 1 && p.K >= 32 && p.C >= 64 &&
        (p.C & 3) == 0 && (p.sxn & 3) == 0 && (p.sxh & 3) == 0 && (p.sxw & 3) == 0 && (p.swk & 3) == 0 && (p.swr & 3) == 0 &&
        (p.sws & 3) == 0 && (long long)p.N * p.Ho * p.Wo >= 1024 && !get_flag("no_f32_mfma") && !get_flag("force_generic") && !get_flag("no_f32_lds")) {
        const long long M = (long long)p.N * p.Ho * p.Wo;
        constexpr int SMEM = 2 * (64 + 256) * 36 * 4;
        set_kernel_name("conv_f32_lds_mfma");
        static LdsAttrSite attr;
        MV_HIP(attr.ensure((const void*)conv_f32_lds_kernel, SMEM));
        hipLaunchKernelGGL(conv_f32_lds_kernel, dim3((unsigned)((M + 255) / 256), (unsigned)((p.K + 63) / 64)), dim3(512), SMEM, st,
                           (const float*)x, (const float*)w, scale, shift, (const float*)residual, (float*)y, pos, p);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    if (x_dtype == MV_F32 && w_dtype == MV_F32 && y_dtype == MV_F32 && p.groups == 1 && p.K >= 8 && !get_flag("no_f32_mfma") && !get_flag("force_generic")) {
        const long long M = (long long)p.N * p.Ho * p.Wo;
        set_kernel_name("conv_f32_mfma");
        hipLaunchKernelGGL(conv_f32_mfma_kernel, dim3((unsigned)((M + 127) / 128), (unsigned)((p.K + 31) / 32)), dim3(256), 0, st,
                           (const float*)x, (const float*)w, scale, shift, (const float*)residual, (float*)y, pos, p);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    set_kernel_name("conv_generic");
#define GO(TX, TW, TY) return conv_generic_launch<TX, TW, TY>(x, w, scale, shift, residual, y, pos, p, st)
    if (x_dtype == MV_F32 && w_dtype == MV_F32 && y_dtype == MV_F32) GO(float, float, float);
    if (x_dtype == MV_BF16 && w_dtype == MV_BF16 && y_dtype == MV_BF16) GO(bf16_t, bf16_t, bf16_t);
    if (x_dtype == MV_BF16 && w_dtype == MV_BF16 && y_dtype == MV_F32) GO(bf16_t, bf16_t, float);
    if (x_dtype == MV_F32 && w_dtype == MV_BF16 && y_dtype == MV_BF16) GO(float, bf16_t, bf16_t);
    if (x_dtype == MV_F32 && w_dtype == MV_F32 && y_dtype == MV_BF16) GO(float, float, bf16_t);
#undef GO
    set_error("conv: unsupported dtype combination x=%d w=%d y=%d", x_dtype, w_dtype, y_dtype);
    return MV_E_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------
// attention, one wave per (b, h, query)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename T>
__global__ void mha_generic_kernel(const T* __restrict__ qkv, T* __restrict__ out, float* __restrict__ probs, int B,
                                   int N, int H, int dh, float scale) {
    extern __shared__ float sm[];  // [N] scores + [dh] q
    float* sc = sm;
    float* qs = sm + N;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x;
    const long long rs = 3LL * H * dh;  // row stride of qkv
    const T* qrow = qkv + ((long long)b * N + i) * rs + h * dh;
    for (int d = lane; d < dh; d += 64) qs[d] = io<T>::ld(qrow + d);
    __syncthreads();
    float mx = -INFINITY;
    for (int j = lane; j < N; j += 64) {
        const T* krow = qkv + ((long long)b * N + j) * rs + (long long)H * dh + h * dh;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s = fmaf(qs[d], io<T>::ld(krow + d), s);
        s *= scale;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < N; j += 64) {
        float e = __expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.f / sum;
    if (probs) {
        float* pr = probs + (((long long)b * H + h) * N + i) * N;
        for (int j = lane; j < N; j += 64) pr[j] = sc[j] * inv;
    }
    for (int d = lane; d < dh; d += 64) {
        float o = 0.f;
        for (int j = 0; j < N; ++j) {
            const T* vrow = qkv + ((long long)b * N + j) * rs + 2LL * H * dh + h * dh;
            o = fmaf(sc[j], io<T>::ld(vrow + d), o);
        }
        io<T>::st(out + ((long long)b * N + i) * H * dh + h * dh + d, o * inv);
    }
}

// ------------------------------------------------------------------------------------------
// Swin shifted-window attention core (swin.py:123-250), one wave per (query token, head)
// qkv NHWC [B,Hf,Wf,3C] with channel order [q|k|v][head][dh]; roll / partition / reverse / mask
// are index arithmetic.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void swin_attn_generic_kernel(const T* __restrict__ qkv, const float* __restrict__ bias,
                                         T* __restrict__ out, int B, int Hf, int Wf, int C, int heads, int wsh,
                                         int wsw, int shh, int shw) {
    extern __shared__ float sm[];  // [n] probs + [dh] q
    const int n = wsh * wsw;
    const int dh = C / heads;
    float* sc = sm;
    float* qs = sm + n;
    const int nWw = Wf / wsw, nWh = Hf / wsh;
    int t = blockIdx.x;                  // token index within window-major order
    const int h = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x;
    const int win = t / n, ti = t % n;
    const int wy = win / nWw, wx = win % nWw;
    (void)nWh;
    // rolled coordinates of the query
    const int qy = wy * wsh + ti / wsw, qx = wx * wsw + ti % wsw;
    const int oy = (qy + shh) % Hf, ox = (qx + shw) % Wf;  // np.roll(x,-s)[i] = x[(i+s)%n]
    const long long rs = 3LL * C;
    const T* qrow = qkv + (((long long)b * Hf + oy) * Wf + ox) * rs + h * dh;
    const float qscale = rsqrtf((float)dh);
    for (int d = lane; d < dh; d += 64) qs[d] = io<T>::ld(qrow + d) * qscale;
    __syncthreads();
    const bool shifted = (shh + shw) > 0;
    auto region = [&](int y, int x) {
        int rh = (y < Hf - wsh) ? 0 : (y < Hf - shh ? 1 : 2);
        int rw = (x < Wf - wsw) ? 0 : (x < Wf - shw ? 1 : 2);
        return rh * 3 + rw;
    };
    const int qreg = region(qy, qx);
    float mx = -INFINITY;
    for (int j = lane; j < n; j += 64) {
        const int ky = wy * wsh + j / wsw, kx = wx * wsw + j % wsw;
        const int yy = (ky + shh) % Hf, xx = (kx + shw) % Wf;
        const T* krow = qkv + (((long long)b * Hf + yy) * Wf + xx) * rs + C + h * dh;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s = fmaf(qs[d], io<T>::ld(krow + d), s);
        s += bias[((long long)h * n + ti) * n + j];
        if (shifted && region(ky, kx) != qreg) s += -100.0f;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n; j += 64) {
        float e = __expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    __syncthreads();
    const float inv = 1.f / sum;
    for (int d = lane; d < dh; d += 64) {
        float o = 0.f;
        for (int j = 0; j < n; ++j) {
            const int ky = wy * wsh + j / wsw, kx = wx * wsw + j % wsw;
            const int yy = (ky + shh) % Hf, xx = (kx + shw) % Wf;
            const T* vrow = qkv + (((long long)b * Hf + yy) * Wf + xx) * rs + 2 * C + h * dh;
            o = fmaf(sc[j], io<T>::ld(vrow + d), o);
        }
        io<T>::st(out + (((long long)b * Hf + oy) * Wf + ox) * C + h * dh + d, o * inv);
    }
}

// ------------------------------------------------------------------------------------------
// fp32 attention with the keys and values of one (image, head [, window]) in LDS: the generic kernels above re-read every K and V row
// from global memory for every query (one wave per query); here a block stages them once (rows padded to dh + 1 floats:
// conflict-free column walks) and its four waves share them.  SWIN: the tokens of a window, gathered through the cyclic shift,
// relative-position bias + region mask (swin.py:90-255); else the N tokens of an image, optional probabilities out (vit.py:64-73).
// The fp32 compute mode and the training step's forward run here.
// ------------------------------------------------------------------------------------------
struct AttnF32P {
    int n, dh, H, C;                 // tokens per group, head width, heads, channels (H * dh)
    int Hf, Wf, wsh, wsw, shh, shw;  // SWIN only
    float scale;
};
template <bool SWIN>
__global__ __launch_bounds__(SWIN ? 256 : 1024) void attn_f32_lds_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                           float* __restrict__ out, float* __restrict__ probs, const AttnF32P p) {
    extern __shared__ float sm[];
    const int n = p.n, dh = p.dh, DP = dh + 1, C = p.C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = SWIN ? 256 : 1024, NW = NT / 64;     // an image's 197 x 64 keys and values fill most of the LDS: one block per CU, so
    float* Ks = sm;                                     // it brings 16 waves; a window's 49 x 32 leave room for several 4-wave blocks
    float* Vs = Ks + n * DP;
    float* sc = Vs + n * DP + wave * (n + dh);          // this wave's scores, then its query
    float* qs = sc + n;
    const int grp = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int nWw = SWIN ? p.Wf / p.wsw : 1;
    const int wy = SWIN ? grp / nWw : 0, wx = SWIN ? grp - wy * nWw : 0;
    auto tok_row = [&](int j) -> long long {           // token j of the group -> its row in qkv / out
        if (!SWIN) return (long long)b * n + j;
        const int ty = j / p.wsw, tx = j - ty * p.wsw;
        const int yy = (wy * p.wsh + ty + p.shh) % p.Hf, xx = (wx * p.wsw + tx + p.shw) % p.Wf;     // np.roll(x, -s)[i] = x[(i + s) % n]
        return ((long long)b * p.Hf + yy) * p.Wf + xx;
    };
    auto region = [&](int j) -> int {                  // shift-mask region of token j (rolled coordinates, swin.py:190-209)
        const int ty = j / p.wsw, tx = j - ty * p.wsw;
        const int y = wy * p.wsh + ty, x = wx * p.wsw + tx;
        const int rh = (y < p.Hf - p.wsh) ? 0 : (y < p.Hf - p.shh ? 1 : 2);
        const int rw = (x < p.Wf - p.wsw) ? 0 : (x < p.Wf - p.shw ? 1 : 2);
        return rh * 3 + rw;
    };
    const long long rs = 3LL * C;
    for (int idx = tid; idx < n * dh; idx += NT) {
        const int j = idx / dh, d = idx - j * dh;
        const float* r = qkv + tok_row(j) * rs + h * dh + d;
        Ks[j * DP + d] = r[C];
        Vs[j * DP + d] = r[2 * C];
    }
    __syncthreads();
    const bool shifted = SWIN && (p.shh + p.shw) > 0;
    for (int i = wave; i < n; i += NW) {
        const long long qrow = tok_row(i);
        for (int d = lane; d < dh; d += 64) qs[d] = qkv[qrow * rs + h * dh + d] * p.scale;
        wave_lds_fence();
        const int qreg = shifted ? region(i) : 0;
        float mx = -INFINITY;
        for (int j = lane; j < n; j += 64) {
            float s = 0.f;
            for (int d = 0; d < dh; ++d) s = fmaf(qs[d], Ks[j * DP + d], s);
            if (SWIN) {
                s += bias[((long long)h * n + i) * n + j];
                if (shifted && region(j) != qreg) s += -100.0f;
            }
            sc[j] = s;
            mx = fmaxf(mx, s);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < n; j += 64) {
            const float e = __expf(sc[j] - mx);
            sc[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        wave_lds_fence();
        if (!SWIN && probs)
            for (int j = lane; j < n; j += 64) probs[(((long long)b * p.H + h) * n + i) * n + j] = sc[j] * inv;
        for (int d = lane; d < dh; d += 64) {
            float o = 0.f;
            for (int j = 0; j < n; ++j) o = fmaf(sc[j], Vs[j * DP + d], o);
            out[qrow * C + h * dh + d] = o * inv;
        }
        wave_lds_fence();                               // before the next query overwrites this wave's scores
    }
}
template <bool SWIN>
static int attn_f32_lds_go(const float* qkv, const float* bias, float* out, float* probs, const AttnF32P& p, int groups, int B,
                           hipStream_t st) {
    constexpr int NW = SWIN ? 4 : 16;
    const size_t smem = ((size_t)2 * p.n * (p.dh + 1) + NW * (size_t)(p.n + p.dh)) * sizeof(float);
    auto kern = attn_f32_lds_kernel<SWIN>;
    static LdsAttrSite attr;
    MV_HIP(attr.ensure((const void*)kern, smem));
    hipLaunchKernelGGL(kern, dim3((unsigned)groups, (unsigned)p.H, (unsigned)B), dim3(NW * 64), smem, st, qkv, bias, out, probs, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}
static bool attn_f32_lds_fits(int n, int dh) {
    return ((size_t)2 * n * (dh + 1) + 16 * (size_t)(n + dh)) * sizeof(float) <= 150 * 1024 && !get_flag("no_attn_f32_lds");
}

// ------------------------------------------------------------------------------------------
// memory-bound ops
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool_nhwc_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho,
                                    int Wo, int kh, int kw, int sh, int sw, int ph, int pw) {
    const long long total = (long long)N * Ho * Wo * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long m = i / C;
        const int wo = (int)(m % Wo);
        m /= Wo;
        const int ho = (int)(m % Ho);
        const int n = (int)(m / Ho);
        float best = -INFINITY;
        for (int r = 0; r < kh; ++r) {
            const int hi = ho * sh - ph + r;
            if (hi < 0 || hi >= H) continue;
            for (int s = 0; s < kw; ++s) {
                const int wi = wo * sw - pw + s;
                if (wi < 0 || wi >= W) continue;
                best = fmaxf(best, io<T>::ld(x + (((long long)n * H + hi) * W + wi) * C + c));
            }
        }
        io<T>::st(y + i, best);
    }
}

// 8 bf16 channels per thread (16-byte loads/stores), C % 8 == 0
__global__ void maxpool_nhwc_bf16x8_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int N, int H, int W,
                                           int C8, int Ho, int Wo, int kh, int kw, int sh, int sw, int ph, int pw) {
    const long long total = (long long)N * Ho * Wo * C8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8);
        long long m = i / C8;
        const int wo = (int)(m % Wo);
        m /= Wo;
        const int ho = (int)(m % Ho);
        const int n = (int)(m / Ho);
        float best[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
        for (int r = 0; r < kh; ++r) {
            const int hi = ho * sh - ph + r;
            if (hi < 0 || hi >= H) continue;
            for (int s = 0; s < kw; ++s) {
                const int wi = wo * sw - pw + s;
                if (wi < 0 || wi >= W) continue;
                const uint4 v = x[(((long long)n * H + hi) * W + wi) * C8 + c];
                const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    best[2 * e] = fmaxf(best[2 * e], __uint_as_float(u[e] << 16));
                    best[2 * e + 1] = fmaxf(best[2 * e + 1], __uint_as_float(u[e] & 0xffff0000u));
                }
            }
        }
        uint4 o;
        // inputs are exact bf16 values, max is exact: truncation == rounding
        o.x = (__float_as_uint(best[0]) >> 16) | (__float_as_uint(best[1]) & 0xffff0000u);
        o.y = (__float_as_uint(best[2]) >> 16) | (__float_as_uint(best[3]) & 0xffff0000u);
        o.z = (__float_as_uint(best[4]) >> 16) | (__float_as_uint(best[5]) & 0xffff0000u);
        o.w = (__float_as_uint(best[6]) >> 16) | (__float_as_uint(best[7]) & 0xffff0000u);
        y[i] = o;
    }
}

__device__ __forceinline__ void adaptive_bounds(int n, int t, int i, int& lo, int& hi) {
    // equinox AdaptiveAvgPool rule
    if (n % t == 0) {
        const int k = n / t;
        lo = i * k;
        hi = lo + k;
    } else {
        const int big = n % t, k = n / t;
        if (i < big) {
            lo = i * (k + 1);
            hi = lo + k + 1;
        } else {
            lo = big * (k + 1) + (i - big) * k;
            hi = lo + k;
        }
    }
}

template <typename TI, typename TO>
__global__ void adaptive_avgpool_nhwc_kernel(const TI* __restrict__ x, TO* __restrict__ y, int N, int H, int W,
                                             int C, int oh, int ow) {
    const long long total = (long long)N * oh * ow * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long m = i / C;
        const int j = (int)(m % ow);
        m /= ow;
        const int ii = (int)(m % oh);
        const int n = (int)(m / oh);
        int h0, h1, w0, w1;
        adaptive_bounds(H, oh, ii, h0, h1);
        adaptive_bounds(W, ow, j, w0, w1);
        float s = 0.f;
        for (int hh = h0; hh < h1; ++hh)
            for (int ww = w0; ww < w1; ++ww) s += io<TI>::ld(x + (((long long)n * H + hh) * W + ww) * C + c);
        io<TO>::st(y + i, s / (float)((h1 - h0) * (w1 - w0)));
    }
}

// Global average pool (oh = ow = 1) of a bf16 NHWC map, 8 channels (16 bytes) per thread, pixels unrolled 4-wide
// so several loads are in flight (the element-per-thread kernel above is latency-bound: 10 us for 12.8 MB).
template <typename TO>
__global__ void global_avgpool_bf16x8_kernel(const uint4* __restrict__ x, TO* __restrict__ y, int N, int HW, int C8) {
    const long long total = (long long)N * C8;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const int n = (int)(i / C8);
        const uint4* xp = x + (long long)n * HW * C8 + c8;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int p = 0;
        for (; p + 4 <= HW; p += 4) {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = xp[(long long)(p + j) * C8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    s[2 * e] += __uint_as_float(w[e] << 16);
                    s[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                }
            }
        }
        for (; p < HW; ++p) {
            const uint4 v = xp[(long long)p * C8];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[2 * e] += __uint_as_float(w[e] << 16);
                s[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
            }
        }
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < 8; ++e) io<TO>::st(y + i * 8 + e, s[e] * inv);
    }
}

// The same reduction for LARGE maps (squeeze-excitation on 112 x 112 ... 14 x 14 maps, layers/squeeze.py:56; DeepLab's pooled
// branch): one thread per (image, 8 channels) would walk 12 544 pixels serially with a few thousand threads on the whole chip
// (measured 1.27 ms for 128 x 112 x 112 x 32).  Here a block of 256 threads owns one image and up to 32 channel chunks; the
// threads of a chunk stride over the pixels (whole 128-byte lines per pixel across the chunk lanes) and meet in LDS.
template <typename TI, typename TO>
__global__ __launch_bounds__(1024) void global_avgpool_wide_kernel(const TI* __restrict__ x, TO* __restrict__ y, int HW, int C8,
                                                                   int cpb, int pl) {
    __shared__ float red[1024 * 8];
    const int n = blockIdx.x, c8 = blockIdx.y * cpb + (int)threadIdx.x % cpb, lp = (int)threadIdx.x / cpb;
    const long long C = (long long)C8 * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (lp < pl && c8 < C8) {
        const TI* xp = x + (long long)n * HW * C + c8 * 8;
        auto ld = [&](int p, float4& a, float4& b) {
            a = Out4<TI>::ld(xp + (long long)p * C);
            b = Out4<TI>::ld(xp + (long long)p * C + 4);
        };
        auto acc = [&](const float4& a, const float4& b) {
            s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w; s[4] += b.x; s[5] += b.y; s[6] += b.z; s[7] += b.w;
        };
        int p = lp;
        for (; p + 3 * pl < HW; p += 4 * pl) {              // four pixels in flight per thread (1024 threads per image: the block
            float4 a[4], b[4];                              // count is only images x channel chunks)
#pragma unroll
            for (int u = 0; u < 4; ++u) ld(p + u * pl, a[u], b[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc(a[u], b[u]);
        }
        for (; p < HW; p += pl) {
            float4 a, b;
            ld(p, a, b);
            acc(a, b);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
    __syncthreads();
    if (lp == 0 && c8 < C8) {
        for (int q = 1; q < pl; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += red[(q * cpb + (int)threadIdx.x) * 8 + e];
        const float inv = 1.f / (float)HW;
#pragma unroll
        for (int e = 0; e < 8; ++e) io<TO>::st(y + ((long long)n * C8 + c8) * 8 + e, s[e] * inv);
    }
}

// LayerNorm: one wave per row, two-pass in registers/LDS-free (row re-read from L1/L2).
template <typename TI, typename TO>
__global__ void layernorm_kernel(const TI* __restrict__ x, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, TO* __restrict__ y, long long M, int C,
                                 long long xs, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const TI* xr = x + row * xs;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += io<TI>::ld(xr + c);
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = io<TI>::ld(xr + c) - mean;
        v += d * d;
    }
    const float rstd = rsqrtf(wave_sum(v) / (float)C + eps);
    TO* yr = y + row * C;
    for (int c = lane; c < C; c += 64) {
        float o = (io<TI>::ld(xr + c) - mean) * rstd;
        if (gamma) o *= gamma[c];
        if (beta) o += beta[c];
        io<TO>::st(yr + c, o);
    }
}

// Vectorised LayerNorm: one wave per row, the whole row held in registers, 16-byte loads.
// TI = bf16 (8 elements per 16-byte chunk) or fp32 (4 elements per chunk, the fp32 residual stream);
// TO = bf16 or fp32.  CHUNKS = 16-byte chunks per lane (row length <= 64 * CHUNKS chunks).
template <typename T> struct Vec16;
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void ld(const bf16_t* p, float* v) {
        const uint4 u = *(const uint4*)p;
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = __uint_as_float(w[e] << 16);
            v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
        }
    }
};
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void ld(const float* p, float* v) {
        const float4 u = *(const float4*)p;
        v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
    }
};
template <typename TO, int N> struct StN;
template <> struct StN<bf16_t, 8> {
    __device__ static __forceinline__ void st(bf16_t* p, const float* o) {
        *(uint4*)p = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
    }
};
template <> struct StN<bf16_t, 4> {
    __device__ static __forceinline__ void st(bf16_t* p, const float* o) {
        *(uint2*)p = make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
    }
};
template <> struct StN<float, 8> {
    __device__ static __forceinline__ void st(float* p, const float* o) {
        *(float4*)p = make_float4(o[0], o[1], o[2], o[3]);
        *(float4*)(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
};
template <> struct StN<float, 4> {
    __device__ static __forceinline__ void st(float* p, const float* o) { *(float4*)p = make_float4(o[0], o[1], o[2], o[3]); }
};

// WIDTH lanes cooperate on one row (64 / WIDTH rows per wave per step); every wave walks a grid-stride list of
// row groups and issues the NEXT group's loads before reducing the current one, so the row latency is hidden
// by the wave itself (the one-row-per-short-lived-wave version ran at 2.5 TB/s on [50432 x 768] fp32).
template <int WIDTH> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <typename TI, typename TO, int CHUNKS, int WIDTH>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const TI* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, TO* __restrict__ y, long long M,
                                                            int C, long long xs, float eps) {
    constexpr int N = Vec16<TI>::N;
    constexpr int RPW = 64 / WIDTH;                      // rows per wave per step
    const int lane = threadIdx.x & 63;
    const int sub = lane / WIDTH, sl = lane % WIDTH;     // which row of the group, lane within the row
    const long long gw = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * (blockDim.x >> 6);
    const int nch = C / N;
    const float invC = 1.0f / (float)C;

    // gamma / beta live in registers for the whole kernel (per-row re-loads through null checks were compiled
    // into 8 serialized scalar loads per chunk); chunk indices are clamped so every load is unconditional.
    float gm[CHUNKS][N], bt[CHUNKS][N];
    int cc[CHUNKS];
#pragma unroll
    for (int i = 0; i < CHUNKS; ++i) {
        const int c = sl + WIDTH * i;
        cc[i] = c < nch ? c : nch - 1;
#pragma unroll
        for (int e = 0; e < N; e += 4) {
            float4 gv = make_float4(1.f, 1.f, 1.f, 1.f), bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gamma) gv = *(const float4*)(gamma + cc[i] * N + e);
            if (beta) bv = *(const float4*)(beta + cc[i] * N + e);
            gm[i][e] = gv.x; gm[i][e + 1] = gv.y; gm[i][e + 2] = gv.z; gm[i][e + 3] = gv.w;
            bt[i][e] = bv.x; bt[i][e + 1] = bv.y; bt[i][e + 2] = bv.z; bt[i][e + 3] = bv.w;
        }
    }

    float cur[CHUNKS][N], nxt[CHUNKS][N];
    const long long groups = (M + RPW - 1) / RPW;
    auto load = [&](float (*v)[N], long long group) {
        if (group >= groups) group = groups - 1;          // clamped re-read instead of a conditional load
        long long row = group * RPW + sub;
        if (row >= M) row = M - 1;
        const TI* xr = x + row * xs;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) Vec16<TI>::ld(xr + cc[i] * N, v[i]);
    };
    // One step: issue the loads of the group after `gg` into `b`, then reduce / normalise / store `a`.  The
    // two register sets swap roles every step (loop unrolled by two) so nothing is copied and the wave only
    // ever waits for the OLDER of two row groups in flight.
    auto step = [&](float (*a)[N], float (*b)[N], long long gg) {
        load(b, gg + nw);
        asm volatile("" : "+v"(a[0][0]) : : "memory");  // keep the prefetch ahead of the arithmetic on `a`
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const bool on = sl + WIDTH * i < nch;
#pragma unroll
            for (int e = 0; e < N; ++e) s += on ? a[i][e] : 0.f;
        }
        const float mean = group_sum<WIDTH>(s) * invC;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const bool on = sl + WIDTH * i < nch;
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const float d = a[i][e] - mean;
                q += on ? d * d : 0.f;
            }
        }
        const float rstd = rsqrtf(group_sum<WIDTH>(q) * invC + eps);
        const long long row = gg * RPW + sub;
        const bool live = gg < groups && row < M;
        TO* yr = y + row * C;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            float o[N];
#pragma unroll
            for (int e = 0; e < N; ++e) o[e] = (a[i][e] - mean) * rstd * gm[i][e] + bt[i][e];
            if (live && sl + WIDTH * i < nch) StN<TO, N>::st(yr + cc[i] * N, o);
        }
    };
    long long g = gw;
    if (g >= groups) return;
    load(cur, g);
    for (; g < groups; g += 2 * nw) {
        step(cur, nxt, g);
        step(nxt, cur, g + nw);
    }
}

// The same LayerNorm on <= 64 VGPRs and no LDS (fp32 rows in, bf16 out, C = 256 * CHUNKS <= 1024): one wave per row, the row in
// CHUNKS x 4 registers, no second row in flight.  Alone it is slower than the kernel above; its point is WHERE it can run: a
// Linear-layer igemm8 workgroup (igemm8.hip, LIN) takes 8 waves x 224 VGPRs of a CU, which leaves 64 registers per SIMD lane --
// room for exactly one such wave per SIMD.  With two graph lanes the LayerNorm of one lane then runs UNDER the other lane's GEMM
// main loop (whose memory pipe is idle) instead of time-slicing the CUs with it.
template <int CHUNKS>
__global__ __launch_bounds__(256) void layernorm_slim_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16_t* __restrict__ y, long long M,
                                                             float eps) {
    constexpr int C = 256 * CHUNKS;
    const int lane = threadIdx.x & 63;
    const long long gw = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nw = (long long)gridDim.x * 4;
    for (long long row = gw; row < M; row += nw) {
        const float* xr = x + row * C + lane * 4;
        float4 v[CHUNKS];
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) v[i] = *(const float4*)(xr + 256 * i);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = group_sum<64>(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
            q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
        }
        const float rstd = rsqrtf(group_sum<64>(q) * (1.0f / C) + eps);
        bf16_t* yr = y + row * C + lane * 4;
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i) {
            const float4 g = *(const float4*)(gamma + 256 * i + lane * 4), b = *(const float4*)(beta + 256 * i + lane * 4);
            *(uint2*)(yr + 256 * i) = make_uint2(pack_bf2(v[i].x * rstd * g.x + b.x, v[i].y * rstd * g.y + b.y),
                                                 pack_bf2(v[i].z * rstd * g.z + b.z, v[i].w * rstd * g.w + b.w));
        }
    }
}

template <typename T>
__global__ void eltwise_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, int act) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        io<T>::st(y + i, apply_act_rt(io<T>::ld(x + i), act));
}
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long long n,
                           int act) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        io<T>::st(y + i, apply_act_rt(io<T>::ld(a + i) + io<T>::ld(b + i), act));
}
template <typename T>
__global__ void channel_scale_kernel(const T* __restrict__ x, const T* __restrict__ sc, T* __restrict__ y, long long HW, int C,
                                     long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long b = i / ((long long)C * HW);
        io<T>::st(y + i, io<T>::ld(x + i) * io<T>::ld(sc + b * C + c));
    }
}
// The element-wise passes with 8 values (16 bytes in bf16) per thread: what the SqueezeExcitation multiply, the un-fused activations
// (hard_swish after a kernel that does not fuse it) and the un-fused residual adds of the section-8 f1 families run on.  HBM-bound;
// the one-value-per-thread kernels above reach 1.9 TB/s, these > 5.
template <typename T> __device__ __forceinline__ void ld8(const T* p, float* v) {
    const float4 a = Out4<T>::ld(p), b = Out4<T>::ld(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename T>
__global__ void eltwise_vec8_kernel(const T* __restrict__ x, T* __restrict__ y, long long n8, int act) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        float v[8];
        ld8(x + i * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act_rt(v[e], act);
        Out8<T>::st(y + i * 8, v);
    }
}
template <typename T>
__global__ void add_vec8_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long long n8, int act) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        float v[8], w[8];
        ld8(a + i * 8, v);
        ld8(b + i * 8, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act_rt(v[e] + w[e], act);
        Out8<T>::st(y + i * 8, v);
    }
}
template <typename T>
__global__ void channel_scale_vec8_kernel(const T* __restrict__ x, const T* __restrict__ sc, T* __restrict__ y, long long HW, int C,
                                          long long n8) {
    const long long V = C >> 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const long long pix = i / V;
        const int c = (int)(i - pix * V) * 8;
        const long long b = pix / HW;
        float v[8], w[8];
        ld8(x + i * 8, v);
        ld8(sc + b * C + c, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= w[e];
        Out8<T>::st(y + i * 8, v);
    }
}
static inline int grid_vec8(long long n8) {
    const long long g = (n8 + 255) / 256;
    return (int)(g > 256 * 32 ? 256 * 32 : (g < 1 ? 1 : g));
}

template <typename T>
__global__ void channel_affine_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                      const float* __restrict__ shift, T* __restrict__ y, long long rows, int C,
                                      int act) {
    const long long n = rows * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float v = io<T>::ld(x + i);
        if (scale) v *= scale[c];
        if (shift) v += shift[c];
        io<T>::st(y + i, apply_act_rt(v, act));
    }
}
// 8 consecutive channels per thread (16-byte accesses in bf16), optional residual: y = act(x * scale[c] + shift[c] + residual) --
// the normalisation pass of a training-mode BatchNorm (every convolution output goes through it once: HBM-bound)
template <typename T>
__global__ void channel_affine_vec8_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                           const T* __restrict__ residual, T* __restrict__ y, long long n8, int C, int act) {
    const int V = C >> 3;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)((unsigned long long)i % (unsigned)V) * 8;
        const T* p = x + i * 8;
        const float4 a = Out4<T>::ld(p), b = Out4<T>::ld(p + 4);
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        float4 s0 = make_float4(1.f, 1.f, 1.f, 1.f), s1 = s0, h0 = make_float4(0.f, 0.f, 0.f, 0.f), h1 = h0;
        if (scale) { s0 = *(const float4*)(scale + c); s1 = *(const float4*)(scale + c + 4); }      // c % 8 == 0: 32-byte aligned
        if (shift) { h0 = *(const float4*)(shift + c); h1 = *(const float4*)(shift + c + 4); }
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
        if (residual) {
            const float4 ra = Out4<T>::ld(residual + i * 8), rb = Out4<T>::ld(residual + i * 8 + 4);
            const float r[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += r[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = apply_act_rt(v[e], act);
        Out8<T>::st(y + i * 8, v);
    }
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        io<TO>::st(y + i, io<TI>::ld(x + i));
}

// NCHW <-> NHWC through a 32x33 LDS tile over (C, HW) so both sides stay coalesced.
template <typename TI, typename TO, bool TO_NHWC>
__global__ void layout_kernel(const TI* __restrict__ x, TO* __restrict__ y, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
    const TI* xs = x + (long long)n * C * HW;
    TO* ys = y + (long long)n * C * HW;
    if (TO_NHWC) {  // read [c][p] rows (p contiguous), write [p][c] rows (c contiguous)
        for (int j = ty; j < 32; j += 8) {
            const int c = c0 + j, p = p0 + tx;
            if (c < C && p < HW) tile[j][tx] = io<TI>::ld(xs + (long long)c * HW + p);
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int p = p0 + j, c = c0 + tx;
            if (c < C && p < HW) io<TO>::st(ys + (long long)p * C + c, tile[tx][j]);
        }
    } else {
        for (int j = ty; j < 32; j += 8) {
            const int p = p0 + j, c = c0 + tx;
            if (c < C && p < HW) tile[j][tx] = io<TI>::ld(xs + (long long)p * C + c);
        }
        __syncthreads();
        for (int j = ty; j < 32; j += 8) {
            const int c = c0 + j, p = p0 + tx;
            if (c < C && p < HW) io<TO>::st(ys + (long long)c * HW + p, tile[tx][j]);
        }
    }
}

template <typename T>
__global__ void cls_pos_kernel(const float* __restrict__ cls, const float* __restrict__ pos, T* __restrict__ tok,
                               int B, int tok_stride, int D) {
    const long long n = (long long)B * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int d = (int)(i % D);
        const int b = (int)(i / D);
        io<T>::st(tok + (long long)b * tok_stride * D + d, cls[d] + pos[d]);
    }
}

template <typename T>
__global__ void patch_merge_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long n = (long long)B * Ho * Wo * 4 * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (4 * C));
        long long m = i / (4 * C);
        const int wo = (int)(m % Wo);
        m /= Wo;
        const int ho = (int)(m % Ho);
        const int b = (int)(m / Ho);
        const int q = c4 / C, c = c4 % C;
        // block order [ (0::2,0::2) | (1::2,0::2) | (0::2,1::2) | (1::2,1::2) ]  (swin.py:26-30)
        const int hi = 2 * ho + (q & 1), wi = 2 * wo + (q >> 1);
        float v = 0.f;
        if (hi < H && wi < W) v = io<T>::ld(x + (((long long)b * H + hi) * W + wi) * C + c);
        io<T>::st(y + i, v);
    }
}

// 16-byte chunks: c16 = chunks per source pixel (C * sizeof(T) / 16); y rows are [4 * c16] chunks
__global__ void patch_merge_vec_kernel(const uint4* __restrict__ x, uint4* __restrict__ y, int B, int H, int W, int c16) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long n = (long long)B * Ho * Wo * 4 * c16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (4 * c16));
        long long m = i / (4 * c16);
        const int wo = (int)(m % Wo);
        m /= Wo;
        const int ho = (int)(m % Ho);
        const int b = (int)(m / Ho);
        const int q = c4 / c16, c = c4 - q * c16;
        const int hi = 2 * ho + (q & 1), wi = 2 * wo + (q >> 1);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (hi < H && wi < W) v = x[(((long long)b * H + hi) * W + wi) * c16 + c];
        y[i] = v;
    }
}

// Swin patch merging (swin.py:23-31, 61-65): the 2 x 2 neighbourhood gather and the LayerNorm over its 4 C channels in ONE pass --
// a wave per output row reads the four C-float segments straight from the fp32 map ([x[0::2,0::2], x[1::2,0::2], x[0::2,1::2],
// x[1::2,1::2]] order), two-pass statistics in registers, bf16 (or fp32) rows out; the gathered 4C map is never materialised.
// NV = float4 per lane (4C / 256 rounded up); H, W even.
template <int NV, typename OutT>
__global__ __launch_bounds__(256) void patch_merge_ln_kernel(const float* __restrict__ x, const float* __restrict__ gam,
                                                             const float* __restrict__ bet, OutT* __restrict__ y, int B, int H, int W,
                                                             int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int Ho = H / 2, Wo = W / 2, c4 = C / 4, q4 = 4 * c4;          // float4 per source pixel / per output row
    const long long rows = (long long)B * Ho * Wo;
    const long long w0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    float4 g[NV], bb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int q = lane + 64 * i;
        g[i] = q < q4 ? ((const float4*)gam)[q] : make_float4(0, 0, 0, 0);
        bb[i] = q < q4 ? ((const float4*)bet)[q] : make_float4(0, 0, 0, 0);
    }
    for (long long m = w0; m < rows; m += nw) {
        const int wo = (int)(m % Wo);
        const long long t = m / Wo;
        const int ho = (int)(t % Ho), b = (int)(t / Ho);
        float4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = lane + 64 * i;
            const int seg = q / c4, c = q - seg * c4;                  // segment 0..3 -> pixel (2 ho + (seg & 1), 2 wo + (seg >> 1))
            v[i] = make_float4(0, 0, 0, 0);
            if (q < q4) v[i] = ((const float4*)x)[(((long long)b * H + 2 * ho + (seg & 1)) * W + 2 * wo + (seg >> 1)) * c4 + c];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)(4 * C);
        float qq = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (lane + 64 * i < q4) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                qq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o);
        const float rstd = rsqrtf(qq / (float)(4 * C) + eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = lane + 64 * i;
            if (q < q4) {
                const float4 o = make_float4(fmaf(v[i].x * rstd, g[i].x, bb[i].x), fmaf(v[i].y * rstd, g[i].y, bb[i].y),
                                             fmaf(v[i].z * rstd, g[i].z, bb[i].z), fmaf(v[i].w * rstd, g[i].w, bb[i].w));
                Out4<OutT>::st((void*)(y + m * 4 * C + 4 * q), o);
            }
        }
    }
}

static inline int grid_for(long long n, int block = 256) {
    long long g = (n + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace mv

using namespace mv;

extern "C" {

int mv_conv2d_nhwc_fwd(const void* x, const void* w, const float* scale, const float* shift, const void* residual,
                       void* y, int N, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw,
                       int dh, int dw, int groups, int act, int in_dtype, int out_dtype, mv_stream_t stream) {
    MV_CHECK_ARG(act >= MV_ACT_NONE && act <= MV_ACT_SILU, "conv2d_nhwc: unknown activation %d", act);
    MV_CHECK_ARG(x && w && y, "conv2d_nhwc: NULL pointer");
    MV_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0, "conv2d_nhwc: non-positive dims");
    MV_CHECK_ARG(sh > 0 && sw > 0 && dh > 0 && dw > 0 && ph >= 0 && pw >= 0, "conv2d_nhwc: bad stride/dilation/pad");
    MV_CHECK_ARG(groups > 0 && C % groups == 0 && K % groups == 0, "conv2d_nhwc: C=%d K=%d not divisible by groups=%d",
                 C, K, groups);
    const int Ho = (H + 2 * ph - dh * (R - 1) - 1) / sh + 1;
    const int Wo = (W + 2 * pw - dw * (S - 1) - 1) / sw + 1;
    MV_CHECK_ARG(Ho > 0 && Wo > 0, "conv2d_nhwc: empty output (%d x %d)", Ho, Wo);
    hipStream_t st = (hipStream_t)stream;
    {   // pointwise layers whose reduction is not a multiple of 64 (Swin C = 96) still fit the streaming kernel
        const long long M = (long long)N * Ho * Wo;
        const bool dense1x1 = R == 1 && S == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0 && groups == 1;
        if (!get_flag("force_generic") && !get_flag("no_stream") && !get_flag("igemm_tile") && dense1x1 && C % 64 != 0 &&
            stream1x1_supported(C, K, in_dtype, out_dtype, M))
            return stream1x1_launch(x, w, scale, shift, residual, y, M, C, K, act, out_dtype, st);
    }
    if (!get_flag("force_generic") && igemm_supported(C, K, R, S, groups, in_dtype, out_dtype))
        return igemm_launch(x, w, scale, shift, residual, y, N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, act, in_dtype,
                            out_dtype, st);
    if (!get_flag("force_generic") && !get_flag("no_oddc") && igemm_oddc_supported(C, K, groups, in_dtype, out_dtype) &&
        (long long)N * Ho * Wo < (1LL << 31) - 256)                // widths like 24 / 96 / 144 / 160: zero-filled last k-tile
        return igemm_oddc_launch(x, w, scale, shift, residual, y, N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, act, out_dtype, st);
    ConvP p;
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S; p.Ho = Ho; p.Wo = Wo;
    p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw; p.groups = groups;
    p.sxn = (long long)H * W * C; p.sxh = (long long)W * C; p.sxw = C; p.sxc = 1;
    const int Cg = C / groups;
    p.swk = (long long)R * S * Cg; p.swr = (long long)S * Cg; p.sws = Cg; p.swc = 1;
    p.act = act; p.tok_stride = 0; p.tok_offset = 0;
    return conv_generic_dispatch(x, w, scale, shift, residual, y, nullptr, p, in_dtype, in_dtype, out_dtype, st);
}

int mv_conv2d_nchw_fwd(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int C,
                       int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int act, int x_dtype,
                       int out_dtype, int tok_stride, int tok_offset, const float* pos, mv_stream_t stream) {
    MV_CHECK_ARG(act >= MV_ACT_NONE && act <= MV_ACT_SILU, "conv2d_nchw: unknown activation %d", act);   // every epilogue of this entry has them all
    MV_CHECK_ARG(x && w && y, "conv2d_nchw: NULL pointer");
    MV_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0, "conv2d_nchw: non-positive dims");
    MV_CHECK_ARG(sh > 0 && sw > 0 && ph >= 0 && pw >= 0, "conv2d_nchw: bad stride/pad");
    const int Ho = (H + 2 * ph - (R - 1) - 1) / sh + 1;
    const int Wo = (W + 2 * pw - (S - 1) - 1) / sw + 1;
    MV_CHECK_ARG(Ho > 0 && Wo > 0, "conv2d_nchw: empty output");
    MV_CHECK_ARG(tok_stride == 0 || tok_stride >= tok_offset + Ho * Wo, "conv2d_nchw: tok_stride too small");
    hipStream_t st = (hipStream_t)stream;
    if (!get_flag("force_generic") && stem_supported(C, K, R, S, x_dtype, out_dtype))
        return stem_launch(x, w, scale, shift, y, N, C, H, W, K, R, S, sh, sw, ph, pw, act, x_dtype, out_dtype,
                           tok_stride, tok_offset, pos, st);
    ConvP p;
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S; p.Ho = Ho; p.Wo = Wo;
    p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = 1; p.dw = 1; p.groups = 1;
    p.sxn = (long long)C * H * W; p.sxc = (long long)H * W; p.sxh = W; p.sxw = 1;
    p.swk = (long long)C * R * S; p.swc = (long long)R * S; p.swr = S; p.sws = 1;
    p.act = act; p.tok_stride = tok_stride; p.tok_offset = tok_offset;
    return conv_generic_dispatch(x, w, scale, shift, nullptr, y, pos, p, x_dtype, out_dtype, out_dtype, st);
}

int mv_conv2d_nchw_f32out_supported(int C, int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int x_dtype) {
    return !get_flag("force_generic") && stem_supported(C, K, R, S, x_dtype, MV_BF16) &&
           stem_f32out_supported(C, H, W, K, R, S, sh, sw, ph, pw, x_dtype);
}

int mv_conv2d_nchw_f32out_fwd(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int C,
                              int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int act, int x_dtype,
                              int tok_stride, int tok_offset, const float* pos, mv_stream_t stream) {
    MV_CHECK_ARG(act >= MV_ACT_NONE && act <= MV_ACT_SILU, "conv2d_nchw_f32out: unknown activation %d", act);
    MV_CHECK_ARG(x && w && y, "conv2d_nchw_f32out: NULL pointer");
    MV_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0, "conv2d_nchw_f32out: non-positive dims");
    if (!mv_conv2d_nchw_f32out_supported(C, H, W, K, R, S, sh, sw, ph, pw, x_dtype)) {
        set_error("conv2d_nchw_f32out: unsupported configuration (ask mv_conv2d_nchw_f32out_supported first)");
        return MV_E_UNSUPPORTED;
    }
    const int Ho = (H - R) / sh + 1, Wo = (W - S) / sw + 1;
    MV_CHECK_ARG(tok_stride == 0 || tok_stride >= tok_offset + Ho * Wo, "conv2d_nchw_f32out: tok_stride too small");
    return stem_launch(x, w, scale, shift, y, N, C, H, W, K, R, S, sh, sw, ph, pw, act, x_dtype, MV_F32, tok_stride, tok_offset, pos,
                       (hipStream_t)stream);
}

int mv_stem_conv_pool_supported(int C, int K, int R, int S, int sh, int sw, int ph, int pw, int pool_k, int pool_s,
                                int pool_p, int act, int x_dtype, int out_dtype, int64_t in_elems) {
    return !get_flag("force_generic") && !get_flag("no_stem_pool") &&
           stem_pool_supported(C, K, R, S, sh, sw, ph, pw, pool_k, pool_s, pool_p, act, x_dtype, out_dtype, in_elems);
}

int mv_stem_conv_pool_fwd(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int C,
                          int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int pool_k, int pool_s,
                          int pool_p, int act, int x_dtype, int out_dtype, mv_stream_t stream) {
    MV_CHECK_FUSED_ACT(act, "stem_conv_pool");
    MV_CHECK_ARG(x && w && y, "stem_conv_pool: NULL pointer");
    MV_CHECK_ARG(N > 0 && H > 0 && W > 0, "stem_conv_pool: non-positive dims");
    if (!mv_stem_conv_pool_supported(C, K, R, S, sh, sw, ph, pw, pool_k, pool_s, pool_p, act, x_dtype, out_dtype,
                                     (int64_t)N * C * H * W)) {
        set_error("stem_conv_pool: unsupported configuration (ask mv_stem_conv_pool_supported first)");
        return MV_E_UNSUPPORTED;
    }
    const int Ho = (H + 2 * ph - R) / sh + 1, Wo = (W + 2 * pw - S) / sw + 1;
    MV_CHECK_ARG(Ho >= pool_k - pool_p && Wo >= pool_k - pool_p && Ho + 2 * pool_p >= pool_k && Wo + 2 * pool_p >= pool_k, "stem_conv_pool: image too small");
    return stem_pool_launch(x, w, scale, shift, y, N, H, W, R, x_dtype, (hipStream_t)stream);
}

int mv_conv1x1_chain_supported(int64_t M, int C, int K, int N2, int dtype) {
    return !get_flag("force_generic") && !get_flag("no_stream") &&
           (chain1x1_supported(M, C, K, N2, dtype) || chain_stream_supported(M, C, K, N2, dtype));
}

int mv_conv1x1_chain_fwd(const void* x, const void* w3, const float* scale3, const float* shift3, const void* residual,
                         void* y, const void* w1, const float* scale1, const float* shift1, void* t1, int64_t M, int C,
                         int K, int N2, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && w3 && residual && y && w1 && t1, "conv1x1_chain: NULL pointer");
    if (!mv_conv1x1_chain_supported(M, C, K, N2, dtype)) {
        set_error("conv1x1_chain: unsupported shape M=%lld C=%d K=%d N2=%d (ask mv_conv1x1_chain_supported first)",
                  (long long)M, C, K, N2);
        return MV_E_UNSUPPORTED;
    }
    MV_CHECK_ARG(y != residual && y != x && t1 != y, "conv1x1_chain: y must not alias x / residual / t1");
    if (chain_stream_supported(M, C, K, N2, dtype))        // weights streamed through LDS (layer2's 128 -> 512 -> 128)
        return chain_stream_launch(x, w3, scale3, shift3, residual, y, w1, scale1, shift1, t1, M, (hipStream_t)stream);
    return chain1x1_launch(x, w3, scale3, shift3, residual, y, w1, scale1, shift1, t1, M, N2, (hipStream_t)stream);
}

int mv_conv1x1_chain_sub_supported(int N, int H, int W, int C, int K, int N2, int dtype) {
    return !get_flag("force_generic") && !get_flag("no_stream") && N > 0 && H > 0 && W > 0 && chain1x1_sub_supported(N, H, W, C, K, N2, dtype);
}

int mv_conv1x1_chain_sub_fwd(const void* x, const void* w3, const float* scale3, const float* shift3, const void* residual,
                             void* y_sub, const void* w1, const float* scale1, const float* shift1, void* t1, int N, int H, int W,
                             int C, int K, int N2, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && w3 && residual && y_sub && w1 && t1, "conv1x1_chain_sub: NULL pointer");
    if (!mv_conv1x1_chain_sub_supported(N, H, W, C, K, N2, dtype)) {
        set_error("conv1x1_chain_sub: unsupported shape N=%d H=%d W=%d C=%d K=%d N2=%d (ask mv_conv1x1_chain_sub_supported first)", N, H,
                  W, C, K, N2);
        return MV_E_UNSUPPORTED;
    }
    MV_CHECK_ARG(y_sub != residual && y_sub != x && t1 != y_sub, "conv1x1_chain_sub: y_sub must not alias x / residual / t1");
    return chain1x1_sub_launch(x, w3, scale3, shift3, residual, y_sub, w1, scale1, shift1, t1, N, H, W, (hipStream_t)stream);
}

int mv_conv1x1_chain_res_supported(int N, int H, int W, int C, int K, int N2, int sub, int dtype) {
    return !get_flag("force_generic") && !get_flag("no_stream") && N > 0 && H > 0 && W > 0 && chain_res_supported(N, H, W, C, K, N2, sub, dtype);
}

int mv_conv1x1_chain_res_fwd(const void* t2, const void* residual, const void* wfrag, const void* shifts, void* y, void* t1, int N, int H,
                             int W, int C, int K, int N2, int sub, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(t2 && residual && wfrag && shifts && y && t1, "conv1x1_chain_res: NULL pointer");
    if (!mv_conv1x1_chain_res_supported(N, H, W, C, K, N2, sub, dtype)) {
        set_error("conv1x1_chain_res: unsupported shape N=%d H=%d W=%d C=%d K=%d N2=%d sub=%d (ask mv_conv1x1_chain_res_supported first)", N,
                  H, W, C, K, N2, sub);
        return MV_E_UNSUPPORTED;
    }
    MV_CHECK_ARG(y != residual && y != t2 && t1 != y && t1 != residual && t1 != t2, "conv1x1_chain_res: outputs must not alias inputs");
    return chain_res_launch(t2, residual, wfrag, shifts, y, t1, N, H, W, sub, (hipStream_t)stream);
}

int mv_conv1x1_chain_rc_supported(int64_t M, int C, int K, int N2, int dtype) {
    return !get_flag("force_generic") && !get_flag("no_stream") && chain_rc_supported(M, C, K, N2, dtype);
}

int mv_conv1x1_chain_rc_fwd(const void* t2, const void* t2_prev, const void* x0, const void* wfrag, const void* shifts, void* y,
                            void* t1, int64_t M, int C, int K, int N2, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(t2 && t2_prev && x0 && wfrag && shifts && y && t1, "conv1x1_chain_rc: NULL pointer");
    if (!mv_conv1x1_chain_rc_supported(M, C, K, N2, dtype)) {
        set_error("conv1x1_chain_rc: unsupported shape M=%lld C=%d K=%d N2=%d (ask mv_conv1x1_chain_rc_supported first)", (long long)M, C,
                  K, N2);
        return MV_E_UNSUPPORTED;
    }
    MV_CHECK_ARG(y != t2 && y != t2_prev && y != x0 && t1 != y && t1 != t2 && t1 != t2_prev && t1 != x0,
                 "conv1x1_chain_rc: outputs must not alias inputs");
    return chain_rc_launch(t2, t2_prev, x0, wfrag, shifts, y, t1, M, (hipStream_t)stream);
}

int mv_conv1x1_chain_rc0_fwd(const void* t2, const void* x0, const void* wfrag, const void* shifts, void* t1, int64_t M, int C, int K,
                             int N2, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(t2 && x0 && wfrag && shifts && t1, "conv1x1_chain_rc0: NULL pointer");
    if (!mv_conv1x1_chain_rc_supported(M, C, K, N2, dtype)) {
        set_error("conv1x1_chain_rc0: unsupported shape M=%lld C=%d K=%d N2=%d (ask mv_conv1x1_chain_rc_supported first)", (long long)M, C,
                  K, N2);
        return MV_E_UNSUPPORTED;
    }
    MV_CHECK_ARG(t1 != t2 && t1 != x0, "conv1x1_chain_rc0: t1 must not alias an input");
    return chain_rc0_launch(t2, x0, wfrag, shifts, t1, M, (hipStream_t)stream);
}

int mv_conv1x1_dual_supported(int64_t M, int C1, int C2, int K, int dtype) {
    return !get_flag("force_generic") && !get_flag("no_igemm2") && igemm2_dual_supported(M, C1, C2, K, dtype);
}

int mv_conv1x1_dual_fwd(const void* x, const void* x2, const void* wcat, const float* scale, const float* shift, void* y,
                        int N, int Ho, int Wo, int C1, int H2, int W2, int C2, int stride2, int K, int act, int dtype,
                        mv_stream_t stream) {
    MV_CHECK_FUSED_ACT(act, "conv1x1_dual");
    MV_CHECK_ARG(x && x2 && wcat && y, "conv1x1_dual: NULL pointer");
    MV_CHECK_ARG(N > 0 && Ho > 0 && Wo > 0 && stride2 >= 1 && (Ho - 1) * stride2 < H2 && (Wo - 1) * stride2 < W2,
                 "conv1x1_dual: the strided source does not cover the output map");
    const int64_t M = (int64_t)N * Ho * Wo;
    if (!mv_conv1x1_dual_supported(M, C1, C2, K, dtype)) {
        set_error("conv1x1_dual: unsupported shape M=%lld C1=%d C2=%d K=%d (ask mv_conv1x1_dual_supported first)", (long long)M,
                  C1, C2, K);
        return MV_E_UNSUPPORTED;
    }
    const int ovd = tile_override("ovd", M, C1, C2, K, stride2, 1);          // 10 / 11 / 12 = igemm8 tiles, 3 = igemm2, 0 = the rule
    {
        int t8 = 0;
        if (get_flag("igemm8") >= 2) t8 = get_flag("igemm8") - 1;
        else if (ovd >= 10 && ovd <= 12) t8 = ovd - 9;
        else if (ovd == 0 && (C1 + C2) / 64 >= 8) t8 = igemm8_wanted(M, C1 + C2, K, 1, 1);   // shorter: igemm2 (tuner, ResNet layer2 entry)
        if (t8)
            return igemm8_dual_launch(x, x2, wcat, scale, shift, nullptr, y, N, Ho, Wo, C1, H2, W2, C2, stride2, K, act, MV_BF16, t8,
                                      (hipStream_t)stream);
    }
    return igemm2_dual_launch(x, x2, wcat, scale, shift, y, N, Ho, Wo, C1, H2, W2, C2, stride2, K, act, (hipStream_t)stream);
}

int mv_conv1x1_dual_chain_supported(int64_t M, int C1, int C2, int K, int N2, int dtype) {
    return !get_flag("force_generic") && !get_flag("no_stream") && chain1x1_dual_supported(M, C1, C2, K, N2, dtype);
}

int mv_conv1x1_dual_chain_fwd(const void* x, const void* x2, const void* wcat, const float* scale, const float* shift, void* y,
                              const void* w1, const float* scale1, const float* shift1, void* t1, int64_t M, int C1, int C2,
                              int K, int N2, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && x2 && wcat && w1 && t1, "conv1x1_dual_chain: NULL pointer");       // y may be NULL: the block output is not stored
    if (!mv_conv1x1_dual_chain_supported(M, C1, C2, K, N2, dtype)) {
        set_error("conv1x1_dual_chain: unsupported shape M=%lld C1=%d C2=%d K=%d N2=%d (ask mv_conv1x1_dual_chain_supported)",
                  (long long)M, C1, C2, K, N2);
        return MV_E_UNSUPPORTED;
    }
    return chain1x1_dual_launch(x, x2, wcat, scale, shift, y, w1, scale1, shift1, t1, M, (hipStream_t)stream);
}

int mv_linear_fwd(const void* x, const void* w, const float* scale, const float* shift, const void* residual,
                  void* y, int64_t M, int N, int K, int act, int in_dtype, int out_dtype, mv_stream_t stream) {
    MV_CHECK_FUSED_ACT(act, "linear");
    MV_CHECK_ARG(x && w && y, "linear: NULL pointer");
    MV_CHECK_ARG(M > 0 && N > 0 && K > 0 && M < (1LL << 31), "linear: bad dims M=%lld N=%d K=%d", (long long)M, N, K);
    if (skinny_f32_supported(M, K, N, in_dtype, out_dtype, residual))       // fp32 classifier heads (exact-fp32 MFMA)
        return skinny_f32_launch(x, w, scale, shift, y, M, K, N, act, (hipStream_t)stream);
    // a Linear over M rows is a 1x1 convolution over an M x 1 image
    return mv_conv2d_nhwc_fwd(x, w, scale, shift, residual, y, 1, (int)M, 1, K, N, 1, 1, 1, 1, 0, 0, 1, 1, 1, act,
                              in_dtype, out_dtype, stream);
}

int mv_linear_split_supported(int64_t M, int N, int K, int dtype) {
    return dtype == MV_BF16 && !get_flag("force_generic") && M < (1LL << 31) - 256 &&
           igemm8_supported(M, K, N, 1, 1, 2LL * M * K, 4LL * N * K) && K % 64 == 0;
}

int mv_linear_split_fwd(const void* x, const void* w_hi_lo, const float* scale, const float* shift, const void* residual,
                        void* y, int64_t M, int N, int K, int act, int in_dtype, int out_dtype, mv_stream_t stream) {
    MV_CHECK_FUSED_ACT(act, "linear_split");
    MV_CHECK_ARG(x && w_hi_lo && y, "linear_split: NULL pointer");
    if (!mv_linear_split_supported(M, N, K, in_dtype)) {
        set_error("linear_split: unsupported shape M=%lld N=%d K=%d (ask mv_linear_split_supported first)", (long long)M, N, K);
        return MV_E_UNSUPPORTED;
    }
    int tile = igemm8_wanted(M, 2 * K, N, 1, 1);
    if (tile == 0) tile = N <= 128 ? 3 : 2;
    // the second reduction source IS x: [x | x] . [w_hi | w_lo]^T = x . (w_hi + w_lo)^T, accumulated in fp32
    return igemm8_dual_launch(x, x, w_hi_lo, scale, shift, residual, y, 1, (int)M, 1, K, (int)M, 1, K, 1, N, act, out_dtype, tile,
                              (hipStream_t)stream);
}

int mv_dwconv2d_supported(int C, int K, int groups, int R, int S, int in_dtype, int out_dtype) {
    return !get_flag("force_generic") && !get_flag("no_dwconv") && dwconv_supported(C, K, groups, R, S, in_dtype, out_dtype);
}

int mv_dwconv2d_nhwc_fwd(const void* x, const void* w_rsc, const float* scale, const float* shift, void* y, int N, int H, int W,
                         int C, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw, int act, int in_dtype, int out_dtype,
                         mv_stream_t stream) {
    MV_CHECK_ARG(x && w_rsc && y, "dwconv2d: NULL pointer");
    MV_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && R > 0 && S > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && dh > 0 && dw > 0,
                 "dwconv2d: bad dims");
    const long long Ho = (H + 2 * ph - dh * (R - 1) - 1) / sh + 1, Wo = (W + 2 * pw - dw * (S - 1) - 1) / sw + 1;
    MV_CHECK_ARG(Ho > 0 && Wo > 0, "dwconv2d: empty output");
    if (!mv_dwconv2d_supported(C, C, C, R, S, in_dtype, out_dtype)) {
        set_error("dwconv2d: unsupported configuration C=%d %dx%d (ask mv_dwconv2d_supported first)", C, R, S);
        return MV_E_UNSUPPORTED;
    }
    return dwconv_launch(x, w_rsc, scale, shift, y, N, H, W, C, R, S, sh, sw, ph, pw, dh, dw, act, (hipStream_t)stream);
}

int mv_conv2d_grouped64_supported(int C, int K, int R, int S, int groups, int in_dtype, int out_dtype) {
    return !get_flag("force_generic") && !get_flag("no_grouped64") && igemm_grouped64_supported(C, K, R, S, groups, in_dtype, out_dtype);
}

int mv_conv2d_nhwc_grouped64_fwd(const void* x, const void* w64, const float* scale, const float* shift, const void* residual,
                                 void* y, int N, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh,
                                 int dw, int groups, int act, int in_dtype, int out_dtype, mv_stream_t stream) {
    MV_CHECK_FUSED_ACT(act, "conv2d_grouped64");
    MV_CHECK_ARG(x && w64 && y, "conv2d_grouped64: NULL pointer");
    MV_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && dh > 0 &&
                 dw > 0, "conv2d_grouped64: bad dims");
    const long long Ho = (H + 2 * ph - dh * (R - 1) - 1) / sh + 1, Wo = (W + 2 * pw - dw * (S - 1) - 1) / sw + 1;
    MV_CHECK_ARG(Ho > 0 && Wo > 0 && (long long)N * Ho * Wo < (1LL << 31) - 256, "conv2d_grouped64: bad output size");
    if (!mv_conv2d_grouped64_supported(C, K, R, S, groups, in_dtype, out_dtype)) {
        set_error("conv2d_grouped64: unsupported configuration C=%d K=%d groups=%d (ask mv_conv2d_grouped64_supported first)", C, K,
                  groups);
        return MV_E_UNSUPPORTED;
    }
    return igemm_grouped64_launch(x, w64, scale, shift, residual, y, N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, groups, act, out_dtype,
                                  (hipStream_t)stream);
}

int mv_conv2d_grouped64_window(int C, int groups) {
    return (groups > 0 && C > 0 && C % groups == 0) ? igemm_grouped64_window(C, groups) : 0;
}

int mv_ln_linear_supported(int64_t M, int N, int K, int x_dtype, int out_dtype) {
    return !get_flag("force_generic") && !get_flag("no_stream") && stream1x1_ln_supported(M, K, N, x_dtype, out_dtype);
}

int mv_ln_linear_fwd(const void* x, const void* w, const float* bias, void* y, int64_t M, int N, int K, float eps, int act,
                     int x_dtype, int out_dtype, mv_stream_t stream) {
    MV_CHECK_FUSED_ACT(act, "ln_linear");
    MV_CHECK_ARG(x && w && y, "ln_linear: NULL pointer");
    if (!mv_ln_linear_supported(M, N, K, x_dtype, out_dtype)) {
        set_error("ln_linear: unsupported configuration M=%lld N=%d K=%d (ask mv_ln_linear_supported first)", (long long)M, N, K);
        return MV_E_UNSUPPORTED;
    }
    return stream1x1_ln_launch(x, w, bias, y, M, K, N, eps, act, x_dtype, (hipStream_t)stream);
}

int mv_ln_mlp_supported(int64_t M, int C, int hidden, int x_dtype) {
    return !get_flag("force_generic") && ln_mlp_supported(M, C, hidden, x_dtype);
}

int mv_ln_mlp_fwd(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int64_t M, int C,
                  int hidden, float eps, int x_dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && w1 && b1 && w2 && b2 && y, "ln_mlp: NULL pointer");
    MV_CHECK_ARG(x != y, "ln_mlp: in-place is not supported (rows are re-read for the residual add)");
    if (!mv_ln_mlp_supported(M, C, hidden, x_dtype)) {
        set_error("ln_mlp: unsupported configuration M=%lld C=%d hidden=%d (ask mv_ln_mlp_supported first)", (long long)M, C, hidden);
        return MV_E_UNSUPPORTED;
    }
    return ln_mlp_launch(x, w1, b1, w2, b2, y, M, eps, x_dtype, (hipStream_t)stream);
}

int mv_conv2d_nchw_split_fwd(const void* x, const void* w_hi, const void* w_lo, const float* scale, const float* shift,
                             void* y, int N, int C, int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int act,
                             int x_dtype, int out_dtype, mv_stream_t stream) {
    MV_CHECK_FUSED_ACT(act, "conv2d_nchw_split");
    MV_CHECK_ARG(x && w_hi && w_lo && y, "conv2d_nchw_split: NULL pointer");
    MV_CHECK_ARG(N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0,
                 "conv2d_nchw_split: bad dims");
    MV_CHECK_ARG((H + 2 * ph - R) / sh + 1 > 0 && (W + 2 * pw - S) / sw + 1 > 0, "conv2d_nchw_split: empty output");
    if (get_flag("force_generic") || !stem_supported(C, K, R, S, x_dtype, out_dtype)) {
        set_error("conv2d_nchw_split: unsupported configuration C=%d K=%d %dx%d", C, K, R, S);
        return MV_E_UNSUPPORTED;
    }
    return stem_launch(x, w_hi, scale, shift, y, N, C, H, W, K, R, S, sh, sw, ph, pw, act, x_dtype, out_dtype, 0, 0, nullptr,
                       (hipStream_t)stream, w_lo);
}

int mv_maxpool2d_nhwc_fwd(const void* x, void* y, int N, int H, int W, int C, int kh, int kw, int sh, int sw, int ph,
                          int pw, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0, "maxpool: bad args");
    MV_CHECK_ARG(2 * ph <= kh && 2 * pw <= kw, "maxpool: padding larger than half the window");
    const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    MV_CHECK_ARG(Ho > 0 && Wo > 0, "maxpool: empty output");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MV_BF16 && C % 8 == 0) {
        set_kernel_name("maxpool_nhwc_bf16x8");
        const long long total = (long long)N * Ho * Wo * (C / 8);
        hipLaunchKernelGGL(maxpool_nhwc_bf16x8_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const uint4*)x,
                           (uint4*)y, N, H, W, C / 8, Ho, Wo, kh, kw, sh, sw, ph, pw);
    } else {
        set_kernel_name("maxpool_nhwc");
        const long long total = (long long)N * Ho * Wo * C;
        if (dtype == MV_BF16)
            hipLaunchKernelGGL(maxpool_nhwc_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, st, (const bf16_t*)x,
                               (bf16_t*)y, N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw);
        else
            hipLaunchKernelGGL(maxpool_nhwc_kernel<float>, dim3(grid_for(total)), dim3(256), 0, st, (const float*)x,
                               (float*)y, N, H, W, C, Ho, Wo, kh, kw, sh, sw, ph, pw);
    }
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_adaptive_avgpool2d_nhwc_fwd(const void* x, void* y, int N, int H, int W, int C, int oh, int ow, int in_dtype,
                                   int out_dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && oh > 0 && ow > 0 && oh <= H && ow <= W, "avgpool: bad args");
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)N * oh * ow * C;
    if (oh == 1 && ow == 1 && C % 8 == 0 && !get_flag("force_generic")) {
        // a block per (image, 32-channel-vector chunk): big maps always; small maps too when one thread per (image, 8 channels)
        // would leave most of the chip without a wave (squeeze-excitation on 14 x 14 / 7 x 7 maps at 128 images; Swin's last map in fp32)
        const bool wide = (long long)H * W >= 256 || ((long long)H * W >= 32 && (long long)N * (C / 8) <= 24576);
        if (wide && N <= 65535) {
            const int C8 = C / 8, cpb = C8 < 32 ? C8 : 32, pl = 1024 / cpb;
            set_kernel_name(in_dtype == MV_BF16 ? "global_avgpool_wide_bf16x8" : "global_avgpool_wide_f32x8");
            dim3 g((unsigned)N, (unsigned)((C8 + cpb - 1) / cpb));
#define GOW(TI, TO) hipLaunchKernelGGL((global_avgpool_wide_kernel<TI, TO>), g, dim3(1024), 0, st, (const TI*)x, (TO*)y, H * W, C8, cpb, pl)
            if (in_dtype == MV_BF16 && out_dtype == MV_BF16) GOW(bf16_t, bf16_t);
            else if (in_dtype == MV_BF16) GOW(bf16_t, float);
            else if (out_dtype == MV_BF16) GOW(float, bf16_t);
            else GOW(float, float);
#undef GOW
            MV_LAUNCH_CHECK();
            return MV_OK;
        }
    }
    if (oh == 1 && ow == 1 && in_dtype == MV_BF16 && C % 8 == 0 && !get_flag("force_generic")) {
        set_kernel_name("global_avgpool_bf16x8");
        const long long nt = (long long)N * (C / 8);
        if (out_dtype == MV_BF16)
            hipLaunchKernelGGL(global_avgpool_bf16x8_kernel<bf16_t>, dim3(grid_for(nt, 64)), dim3(64), 0, st, (const uint4*)x,
                               (bf16_t*)y, N, H * W, C / 8);
        else
            hipLaunchKernelGGL(global_avgpool_bf16x8_kernel<float>, dim3(grid_for(nt, 64)), dim3(64), 0, st, (const uint4*)x,
                               (float*)y, N, H * W, C / 8);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    set_kernel_name("adaptive_avgpool_nhwc");
#define GO(TI, TO)                                                                                              \
    hipLaunchKernelGGL((adaptive_avgpool_nhwc_kernel<TI, TO>), dim3(grid_for(total)), dim3(256), 0, st, (const TI*)x, \
                       (TO*)y, N, H, W, C, oh, ow)
    if (in_dtype == MV_BF16 && out_dtype == MV_BF16) GO(bf16_t, bf16_t);
    else if (in_dtype == MV_BF16 && out_dtype == MV_F32) GO(bf16_t, float);
    else if (in_dtype == MV_F32 && out_dtype == MV_F32) GO(float, float);
    else if (in_dtype == MV_F32 && out_dtype == MV_BF16) GO(float, bf16_t);
#undef GO
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, int64_t M, int C,
                     int64_t x_row_stride, float eps, int in_dtype, int out_dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && y && M > 0 && C > 0, "layernorm: bad args");
    const long long xs = x_row_stride ? (long long)x_row_stride : (long long)C;
    MV_CHECK_ARG(xs >= C, "layernorm: row stride %lld < C=%d", xs, C);
    hipStream_t st = (hipStream_t)stream;
    const int rows_per_block = 4;
    dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block)), block(64 * rows_per_block);
    const int epc = in_dtype == MV_BF16 ? 8 : 4;          // elements per 16-byte chunk of the input
    const int nch = C / epc;
    if (!get_flag("no_ln_slim") && in_dtype == MV_F32 && out_dtype == MV_BF16 && gamma && beta && xs == C && C % 256 == 0 && C <= 768 &&
        !get_flag("force_generic")) {
        set_kernel_name("layernorm_slim");
        long long nb = (M + 3) / 4;
        if (nb > 2048) nb = 2048;
        dim3 sgrid((unsigned)nb), sblock(256);
#define GOS(CH) hipLaunchKernelGGL((layernorm_slim_kernel<CH>), sgrid, sblock, 0, st, (const float*)x, gamma, beta, (bf16_t*)y, (long long)M, eps)
        if (C == 256) GOS(1);
        else if (C == 512) GOS(2);
        else if (C == 768) GOS(3);
        else GOS(4);
#undef GOS
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    if (C % epc == 0 && xs % epc == 0 && nch <= 64 * 8 && !get_flag("force_generic")) {
        set_kernel_name("layernorm_vec");
        const int width = nch <= 16 ? 16 : (nch <= 32 ? 32 : 64);
        const int rpw = 64 / width;
        const long long groups = (M + rpw - 1) / rpw;
        long long nb = (groups + 3) / 4;
        if (nb > 256 * 8) nb = 256 * 8;
        dim3 vgrid((unsigned)nb), vblock(256);
#define GO4(TI, TO, CH, WD) \
    hipLaunchKernelGGL((layernorm_vec_kernel<TI, TO, CH, WD>), vgrid, vblock, 0, st, (const TI*)x, gamma, beta, (TO*)y, \
                       (long long)M, C, xs, eps)
#define GO2(TI, TO)                                   \
    do {                                              \
        if (width == 16) GO4(TI, TO, 1, 16);          \
        else if (width == 32) GO4(TI, TO, 1, 32);     \
        else if (nch <= 64) GO4(TI, TO, 1, 64);       \
        else if (nch <= 128) GO4(TI, TO, 2, 64);      \
        else if (nch <= 192) GO4(TI, TO, 3, 64);      \
        else if (nch <= 256) GO4(TI, TO, 4, 64);      \
        else if (nch <= 384) GO4(TI, TO, 6, 64);      \
        else GO4(TI, TO, 8, 64);                      \
    } while (0)
        if (in_dtype == MV_BF16 && out_dtype == MV_BF16) GO2(bf16_t, bf16_t);
        else if (in_dtype == MV_BF16 && out_dtype == MV_F32) GO2(bf16_t, float);
        else if (in_dtype == MV_F32 && out_dtype == MV_F32) GO2(float, float);
        else GO2(float, bf16_t);
#undef GO2
#undef GO4
    } else {
        set_kernel_name("layernorm");
#define GO(TI, TO)                                                                                         \
    hipLaunchKernelGGL((layernorm_kernel<TI, TO>), grid, block, 0, st, (const TI*)x, gamma, beta, (TO*)y, \
                       (long long)M, C, xs, eps)
        if (in_dtype == MV_BF16 && out_dtype == MV_BF16) GO(bf16_t, bf16_t);
        else if (in_dtype == MV_BF16 && out_dtype == MV_F32) GO(bf16_t, float);
        else if (in_dtype == MV_F32 && out_dtype == MV_F32) GO(float, float);
        else if (in_dtype == MV_F32 && out_dtype == MV_BF16) GO(float, bf16_t);
#undef GO
    }
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_mha_fwd(const void* qkv, void* out, float* probs, int B, int N, int H, int dh, float scale, int dtype,
               mv_stream_t stream) {
    MV_CHECK_ARG(qkv && out && B > 0 && N > 0 && H > 0 && dh > 0, "mha: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (!get_flag("force_generic") && dtype == MV_BF16 && mha_mfma_supported(N, dh, dtype))
        return mha_mfma_launch(qkv, 0, out, probs, B, N, H, dh, scale, nullptr, 1.f, st);
    MV_CHECK_ARG(H <= 65535 && B <= 65535, "mha: H/B too large for the generic kernel");
    if (dtype == MV_F32 && !get_flag("force_generic") && attn_f32_lds_fits(N, dh) && (long long)B * H >= 128) {   // one block per (image, head):
        AttnF32P ap;                                                                                              // fewer do not fill the chip
        memset(&ap, 0, sizeof(ap));
        ap.n = N; ap.dh = dh; ap.H = H; ap.C = H * dh; ap.scale = scale;
        set_kernel_name("mha_f32_lds");
        return attn_f32_lds_go<false>((const float*)qkv, nullptr, (float*)out, probs, ap, 1, B, st);
    }
    const size_t smem = (size_t)(N + dh) * sizeof(float);
    MV_CHECK_ARG(smem <= 64 * 1024, "mha: sequence too long for the generic kernel (N=%d)", N);
    set_kernel_name("mha_generic");
    if (dtype == MV_BF16)
        hipLaunchKernelGGL(mha_generic_kernel<bf16_t>, dim3(N, H, B), dim3(64), smem, st, (const bf16_t*)qkv,
                           (bf16_t*)out, probs, B, N, H, dh, scale);
    else
        hipLaunchKernelGGL(mha_generic_kernel<float>, dim3(N, H, B), dim3(64), smem, st, (const float*)qkv,
                           (float*)out, probs, B, N, H, dh, scale);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_linear_heads_supported(int64_t M, int N, int K, int tokens, int dh, int dtype) {
    return dtype == MV_BF16 && dh == 64 && N % 64 == 0 && K % 64 == 0 && tokens > 0 && M > 0 && M % tokens == 0 &&
           M < (1LL << 31) - 256 && igemm2_wanted(M, K, N, 1, 1) && !get_flag("force_generic") && !get_flag("no_igemm2") &&
           mha_mfma_supported(tokens, dh, dtype);
}

int mv_linear_heads_fwd(const void* x, const void* w, const float* scale, const float* shift, void* y, int64_t M, int N,
                        int K, int tokens, int dh, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && w && y, "linear_heads: NULL pointer");
    if (!mv_linear_heads_supported(M, N, K, tokens, dh, dtype)) {
        set_error("linear_heads: unsupported shape M=%lld N=%d K=%d tokens=%d dh=%d (ask mv_linear_heads_supported first)",
                  (long long)M, N, K, tokens, dh);
        return MV_E_UNSUPPORTED;
    }
    const int ov8 = !get_flag("igemm2_tile") ? tile_override("ovh", M, N, K, 1, 1, 1) : -1;
    {
        int t8 = 0;
        if (get_flag("igemm8") >= 2) t8 = get_flag("igemm8") - 1;
        else if (ov8 >= 10 && ov8 <= 12) t8 = ov8 - 9;
        else if (ov8 == 0) t8 = igemm8_wanted(M, K, N, 1, 1);
        if (t8 && igemm8_supported(M, K, N, 1, 1, 2LL * M * K, 2LL * N * K))
            return igemm8_launch(x, w, scale, shift, nullptr, y, 1, (int)M, 1, K, N, 1, 1, 1, 1, 0, 0, 1, 1, MV_ACT_NONE, MV_BF16, tokens,
                                 t8, (hipStream_t)stream);
    }
    if (ov8 == 2 || ov8 == 3) {
        igemm2_force_tile(ov8);
        const int rc = igemm2_launch(x, w, scale, shift, nullptr, y, 1, (int)M, 1, K, N, 1, 1, 1, 1, 0, 0, 1, 1, MV_ACT_NONE,
                                     MV_BF16, 0, tokens, (hipStream_t)stream);
        igemm2_force_tile(0);
        return rc;
    }
    return igemm2_launch(x, w, scale, shift, nullptr, y, 1, (int)M, 1, K, N, 1, 1, 1, 1, 0, 0, 1, 1, MV_ACT_NONE, MV_BF16, 0,
                         tokens, (hipStream_t)stream);
}

// ---- the LayerNorm between two Linears folded into their epilogues (ViT: norm1 -> qkv, norm2 -> fc1) ----
// The pair is only worth it on the 256 x 256 tiles both GEMMs would take anyway (igemm8_wanted == 1 for BOTH shapes is the
// caller's business: ask _supported for each Linear of the pair).
int mv_linear_lnout_supported(int64_t M, int N, int K, int dtype) {
    return dtype == MV_BF16 && !get_flag("force_generic") && !get_flag("no_igemm8") && !get_flag("no_ln_fold") && igemm8_ln_supported(M, N, K) &&
           igemm8_wanted(M, K, N, 1, 1) == 1;
}

int mv_linear_lnout_fwd(const void* x, const void* w, const float* shift, const void* res, const void* res_lo, void* y, void* y_lo,
                        float* stats, int64_t M, int N, int K, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && w && res && y, "linear_lnout: NULL pointer");
    MV_CHECK_ARG((y_lo != nullptr) == (stats != nullptr), "linear_lnout: y_lo and stats go together (planes out) or are both NULL (fp32 rows out)");
    MV_CHECK_ARG(y_lo || res_lo, "linear_lnout: fp32 rows in and out is mv_linear_fwd");
    MV_CHECK_ARG(y != x && y_lo != x, "linear_lnout: an output aliases the operand rows");
    if (!mv_linear_lnout_supported(M, N, K, dtype)) {
        set_error("linear_lnout: unsupported shape M=%lld N=%d K=%d (ask mv_linear_lnout_supported first)", (long long)M, N, K);
        return MV_E_UNSUPPORTED;
    }
    return igemm8_lnout_launch(x, w, shift, res, res_lo, y, y_lo, stats, M, N, K, (hipStream_t)stream);
}

int mv_linear_lnin_supported(int64_t M, int N, int K, int tokens, int dh, int dtype) {
    if (tokens > 0 && !(dh == 64 && M % tokens == 0 && mha_mfma_supported(tokens, dh, dtype))) return 0;
    return dtype == MV_BF16 && !get_flag("force_generic") && !get_flag("no_igemm8") && !get_flag("no_ln_fold") && K <= 768 &&
           igemm8_ln_supported(M, N, K) && igemm8_wanted(M, K, N, 1, 1) == 1;
}

int mv_linear_lnin_fwd(const void* x, const float* stats, const void* w_folded, const float* colsum, const float* shift, void* y,
                       int64_t M, int N, int K, float eps, int act, int tokens, int dh, int dtype, mv_stream_t stream) {
    MV_CHECK_FUSED_ACT(act, "linear_lnin");
    MV_CHECK_ARG(x && stats && w_folded && colsum && y, "linear_lnin: NULL pointer");
    MV_CHECK_ARG(eps > 0.f, "linear_lnin: eps %g", (double)eps);
    if (!mv_linear_lnin_supported(M, N, K, tokens, dh, dtype)) {
        set_error("linear_lnin: unsupported shape M=%lld N=%d K=%d tokens=%d dh=%d (ask mv_linear_lnin_supported first)", (long long)M, N,
                  K, tokens, dh);
        return MV_E_UNSUPPORTED;
    }
    return igemm8_lnin_launch(x, stats, w_folded, colsum, shift, y, M, N, K, eps, act, tokens, (hipStream_t)stream);
}

int mv_mha_heads_fwd(const void* qkv, void* out, float* probs, int B, int N, int H, int dh, float scale, int dtype,
                     mv_stream_t stream) {
    MV_CHECK_ARG(qkv && out && B > 0 && N > 0 && H > 0 && dh > 0, "mha_heads: bad args");
    if (dtype != MV_BF16 || !mha_mfma_supported(N, dh, dtype)) {
        set_error("mha_heads: unsupported N=%d dh=%d dtype=%d", N, dh, dtype);
        return MV_E_UNSUPPORTED;
    }
    return mha_mfma_launch(qkv, 1, out, probs, B, N, H, dh, scale, nullptr, 1.f, (hipStream_t)stream);
}

int mv_mha_dropout_fwd(const void* qkv, int head_major, void* out, float* probs, const uint32_t* keys, float keep_prob,
                       int B, int N, int H, int dh, float scale, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(qkv && out && keys && B > 0 && N > 0 && H > 0 && dh > 0, "mha_dropout: bad args");
    MV_CHECK_ARG(keep_prob > 0.f && keep_prob <= 1.f, "mha_dropout: keep_prob %g outside (0, 1]", (double)keep_prob);
    MV_CHECK_ARG((long long)H * N * N < (1LL << 32), "mha_dropout: more than 2^32 probabilities per sample");
    if (dtype != MV_BF16 || !mha_mfma_supported(N, dh, dtype)) {
        set_error("mha_dropout: unsupported N=%d dh=%d dtype=%d (bf16, dh 32 / 64, N <= 256)", N, dh, dtype);
        return MV_E_UNSUPPORTED;
    }
    return mha_mfma_launch(qkv, head_major, out, probs, B, N, H, dh, scale, keys, keep_prob, (hipStream_t)stream);
}

int mv_swin_window_attn_fwd(const void* qkv, const float* bias, void* out, int B, int Hf, int Wf, int C, int heads,
                            int ws_h, int ws_w, int shift_h, int shift_w, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(qkv && bias && out && B > 0 && Hf > 0 && Wf > 0 && C > 0 && heads > 0, "swin_attn: bad args");
    MV_CHECK_ARG(C % heads == 0, "swin_attn: C %% heads != 0");
    MV_CHECK_ARG(ws_h > 0 && ws_w > 0 && Hf % ws_h == 0 && Wf % ws_w == 0,
                 "swin_attn: feature map %dx%d is not a multiple of the window %dx%d (reference swin.py:782-790)", Hf,
                 Wf, ws_h, ws_w);
    MV_CHECK_ARG(shift_h >= 0 && shift_w >= 0 && shift_h < ws_h && shift_w < ws_w, "swin_attn: bad shift");
    if (ws_h >= Hf) shift_h = 0;  // swin.py:116-120
    if (ws_w >= Wf) shift_w = 0;
    hipStream_t st = (hipStream_t)stream;
    if (!get_flag("force_generic") && swin_mfma_supported(C, heads, ws_h, ws_w, dtype))
        return swin_mfma_launch(qkv, bias, out, B, Hf, Wf, C, heads, ws_h, ws_w, shift_h, shift_w, nullptr, 1.f, st);
    const int n = ws_h * ws_w, dh = C / heads;
    const int tokens = Hf * Wf;
    MV_CHECK_ARG(heads <= 65535 && B <= 65535, "swin_attn: grid too large");
    if (dtype == MV_F32 && !get_flag("force_generic") && attn_f32_lds_fits(n, dh)) {
        AttnF32P ap;
        ap.n = n; ap.dh = dh; ap.H = heads; ap.C = C; ap.Hf = Hf; ap.Wf = Wf; ap.wsh = ws_h; ap.wsw = ws_w; ap.shh = shift_h; ap.shw = shift_w;
        ap.scale = 1.0f / sqrtf((float)dh);
        set_kernel_name("swin_attn_f32_lds");
        return attn_f32_lds_go<true>((const float*)qkv, bias, (float*)out, nullptr, ap, (Hf / ws_h) * (Wf / ws_w), B, st);
    }
    const size_t smem = (size_t)(n + dh) * sizeof(float);
    set_kernel_name("swin_attn_generic");
    if (dtype == MV_BF16)
        hipLaunchKernelGGL(swin_attn_generic_kernel<bf16_t>, dim3(tokens, heads, B), dim3(64), smem, st,
                           (const bf16_t*)qkv, bias, (bf16_t*)out, B, Hf, Wf, C, heads, ws_h, ws_w, shift_h, shift_w);
    else
        hipLaunchKernelGGL(swin_attn_generic_kernel<float>, dim3(tokens, heads, B), dim3(64), smem, st,
                           (const float*)qkv, bias, (float*)out, B, Hf, Wf, C, heads, ws_h, ws_w, shift_h, shift_w);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_swin_window_attn_dropout_fwd(const void* qkv, const float* bias, void* out, const void* keys, float keep_prob, int B,
                                    int Hf, int Wf, int C, int heads, int ws_h, int ws_w, int shift_h, int shift_w, int dtype,
                                    mv_stream_t stream) {
    MV_CHECK_ARG(qkv && bias && out && keys && B > 0 && Hf > 0 && Wf > 0 && C > 0 && heads > 0, "swin_attn_dropout: bad args");
    MV_CHECK_ARG(C % heads == 0, "swin_attn_dropout: C %% heads != 0");
    MV_CHECK_ARG(ws_h > 0 && ws_w > 0 && Hf % ws_h == 0 && Wf % ws_w == 0,
                 "swin_attn_dropout: feature map %dx%d is not a multiple of the window %dx%d (reference swin.py:782-790)", Hf, Wf,
                 ws_h, ws_w);
    MV_CHECK_ARG(shift_h >= 0 && shift_w >= 0 && shift_h < ws_h && shift_w < ws_w, "swin_attn_dropout: bad shift");
    MV_CHECK_ARG(keep_prob > 0.f && keep_prob <= 1.f, "swin_attn_dropout: keep_prob %g outside (0, 1]", (double)keep_prob);
    if (ws_h >= Hf) shift_h = 0;  // swin.py:116-120
    if (ws_w >= Wf) shift_w = 0;
    if (!swin_mfma_supported(C, heads, ws_h, ws_w, dtype)) {
        set_error("swin_attn_dropout: unsupported C=%d heads=%d window %dx%d dtype=%d (bf16, 32 channels per head, <= 64 tokens)", C,
                  heads, ws_h, ws_w, dtype);
        return MV_E_UNSUPPORTED;
    }
    return swin_mfma_launch(qkv, bias, out, B, Hf, Wf, C, heads, ws_h, ws_w, shift_h, shift_w, (const uint32_t*)keys, keep_prob,
                            (hipStream_t)stream);
}

int mv_patch_merge_gather_nhwc(const void* x, void* y, int B, int H, int W, int C, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && C > 0, "patch_merge: bad args");
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * 4 * C;
    const int esz = dtype == MV_BF16 ? 2 : 4;
    if ((C * esz) % 16 == 0 && !get_flag("force_generic")) {
        set_kernel_name("patch_merge_gather_vec");
        const int c16 = C * esz / 16;
        const long long nv = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * 4 * c16;
        hipLaunchKernelGGL(patch_merge_vec_kernel, dim3(grid_for(nv)), dim3(256), 0, st, (const uint4*)x, (uint4*)y, B, H, W,
                           c16);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    set_kernel_name("patch_merge_gather");
    if (dtype == MV_BF16)
        hipLaunchKernelGGL(patch_merge_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)x,
                           (bf16_t*)y, B, H, W, C);
    else
        hipLaunchKernelGGL(patch_merge_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)x, (float*)y,
                           B, H, W, C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_patch_merge_ln_supported(int H, int W, int C, int x_dtype) {
    return !get_flag("no_patch_merge_ln") && x_dtype == MV_F32 && H % 2 == 0 && W % 2 == 0 && C % 4 == 0 && C <= 384 && C >= 16;
}

int mv_patch_merge_ln_fwd(const void* x, const float* gamma, const float* beta, void* y, int B, int H, int W, int C, float eps,
                          int x_dtype, int out_dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && gamma && beta && y && B > 0, "patch_merge_ln: bad args");
    if (!mv_patch_merge_ln_supported(H, W, C, x_dtype)) {
        set_error("mv_patch_merge_ln_fwd: unsupported %dx%dx%d (ask mv_patch_merge_ln_supported first)", H, W, C);
        return MV_E_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)B * (H / 2) * (W / 2);
    long long blocks = (rows + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    const int nv = (C + 63) / 64;                       // float4 per lane: 4C / 4 / 64
    set_kernel_name("patch_merge_ln_f32in");
#define MV_PML(NVV)                                                                                                              \
    do {                                                                                                                         \
        if (out_dtype == MV_BF16)                                                                                                \
            hipLaunchKernelGGL((patch_merge_ln_kernel<NVV, bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, gamma, \
                               beta, (bf16_t*)y, B, H, W, C, eps);                                                               \
        else                                                                                                                     \
            hipLaunchKernelGGL((patch_merge_ln_kernel<NVV, float>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, gamma,  \
                               beta, (float*)y, B, H, W, C, eps);                                                                \
    } while (0)
    if (nv <= 2) MV_PML(2);
    else if (nv <= 3) MV_PML(3);
    else MV_PML(6);
#undef MV_PML
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_vit_cls_pos_fwd(const float* cls, const float* pos, void* tokens, int B, int tok_stride, int D, int dtype,
                       mv_stream_t stream) {
    MV_CHECK_ARG(cls && pos && tokens && B > 0 && tok_stride > 0 && D > 0, "cls_pos: bad args");
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)B * D;
    set_kernel_name("vit_cls_pos");
    if (dtype == MV_BF16)
        hipLaunchKernelGGL(cls_pos_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, cls, pos, (bf16_t*)tokens, B,
                           tok_stride, D);
    else
        hipLaunchKernelGGL(cls_pos_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, cls, pos, (float*)tokens, B,
                           tok_stride, D);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_eltwise_fwd(const void* x, void* y, int64_t n, int act, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && y && n > 0, "eltwise: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (n % 8 == 0 && !get_flag("eltwise_scalar")) {
        set_kernel_name("eltwise_x8");
        if (dtype == MV_BF16)
            hipLaunchKernelGGL(eltwise_vec8_kernel<bf16_t>, dim3(grid_vec8(n / 8)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y,
                               (long long)(n / 8), act);
        else
            hipLaunchKernelGGL(eltwise_vec8_kernel<float>, dim3(grid_vec8(n / 8)), dim3(256), 0, st, (const float*)x, (float*)y,
                               (long long)(n / 8), act);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    set_kernel_name("eltwise");
    if (dtype == MV_BF16)
        hipLaunchKernelGGL(eltwise_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y,
                           (long long)n, act);
    else
        hipLaunchKernelGGL(eltwise_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)x, (float*)y,
                           (long long)n, act);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_channel_scale_nhwc_fwd(const void* x, const void* sc, void* y, int N, int64_t HW, int C, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && sc && y && N > 0 && HW > 0 && C > 0, "channel_scale: bad args");
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)N * HW * C;
    if (C % 8 == 0 && !get_flag("eltwise_scalar")) {
        set_kernel_name("channel_scale_x8");
        if (dtype == MV_BF16)
            hipLaunchKernelGGL(channel_scale_vec8_kernel<bf16_t>, dim3(grid_vec8(n / 8)), dim3(256), 0, st, (const bf16_t*)x,
                               (const bf16_t*)sc, (bf16_t*)y, (long long)HW, C, n / 8);
        else
            hipLaunchKernelGGL(channel_scale_vec8_kernel<float>, dim3(grid_vec8(n / 8)), dim3(256), 0, st, (const float*)x,
                               (const float*)sc, (float*)y, (long long)HW, C, n / 8);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    set_kernel_name("channel_scale");
    if (dtype == MV_BF16)
        hipLaunchKernelGGL(channel_scale_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)sc,
                           (bf16_t*)y, (long long)HW, C, n);
    else
        hipLaunchKernelGGL(channel_scale_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)x, (const float*)sc,
                           (float*)y, (long long)HW, C, n);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_add_fwd(const void* a, const void* b, void* y, int64_t n, int act, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(a && b && y && n > 0, "add: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (n % 8 == 0 && !get_flag("eltwise_scalar")) {
        set_kernel_name("add_x8");
        if (dtype == MV_BF16)
            hipLaunchKernelGGL(add_vec8_kernel<bf16_t>, dim3(grid_vec8(n / 8)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b,
                               (bf16_t*)y, (long long)(n / 8), act);
        else
            hipLaunchKernelGGL(add_vec8_kernel<float>, dim3(grid_vec8(n / 8)), dim3(256), 0, st, (const float*)a, (const float*)b,
                               (float*)y, (long long)(n / 8), act);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    set_kernel_name("add");
    if (dtype == MV_BF16)
        hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b,
                           (bf16_t*)y, (long long)n, act);
    else
        hipLaunchKernelGGL(add_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)a, (const float*)b,
                           (float*)y, (long long)n, act);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

static int channel_affine_go(const void* x, const float* scale, const float* shift, const void* residual, void* y, int64_t rows, int C,
                             int act, int dtype, hipStream_t st) {
    const long long n = (long long)rows * C;
    if (C % 8 == 0 && !get_flag("affine_scalar")) {          // 16-byte accesses; the scalar kernel serves odd widths (and residual-free calls only)
        const long long n8 = n / 8;
        long long g = (n8 + 255) / 256;
        const int grid = (int)(g > 256 * 32 ? 256 * 32 : g);
        set_kernel_name(residual ? "channel_affine_res_x8" : "channel_affine_x8");
        if (dtype == MV_BF16)
            hipLaunchKernelGGL(channel_affine_vec8_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, scale, shift,
                               (const bf16_t*)residual, (bf16_t*)y, n8, C, act);
        else
            hipLaunchKernelGGL(channel_affine_vec8_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, scale, shift,
                               (const float*)residual, (float*)y, n8, C, act);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    if (residual) {
        set_error("channel_affine_res: C=%d is not a multiple of 8", C);
        return MV_E_UNSUPPORTED;
    }
    set_kernel_name("channel_affine");
    if (dtype == MV_BF16)
        hipLaunchKernelGGL(channel_affine_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, st, (const bf16_t*)x, scale,
                           shift, (bf16_t*)y, (long long)rows, C, act);
    else
        hipLaunchKernelGGL(channel_affine_kernel<float>, dim3(grid_for(n)), dim3(256), 0, st, (const float*)x, scale,
                           shift, (float*)y, (long long)rows, C, act);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_channel_affine_fwd(const void* x, const float* scale, const float* shift, void* y, int64_t rows, int C,
                          int act, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && y && rows > 0 && C > 0, "channel_affine: bad args");
    return channel_affine_go(x, scale, shift, nullptr, y, rows, C, act, dtype, (hipStream_t)stream);
}

// y = act(x * scale[c] + shift[c] + residual): BatchNorm's normalisation + the block's identity + ReLU in one pass (the tail of a
// residual block whose BatchNorm is in training mode and therefore not in the convolution's epilogue; resnet.py:155-160)
int mv_channel_affine_res_fwd(const void* x, const float* scale, const float* shift, const void* residual, void* y, int64_t rows, int C,
                              int act, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && y && residual && rows > 0 && C > 0, "channel_affine_res: bad args");
    return channel_affine_go(x, scale, shift, residual, y, rows, C, act, dtype, (hipStream_t)stream);
}

static int layout_launch(const void* x, void* y, int N, int C, int H, int W, int in_dtype, int out_dtype, bool to_nhwc,
                         hipStream_t st) {
    const int HW = H * W;
    dim3 block(32, 8), grid((HW + 31) / 32, (C + 31) / 32, N);
#define GO(TI, TO)                                                                                                   \
    do {                                                                                                             \
        if (to_nhwc)                                                                                                 \
            hipLaunchKernelGGL((layout_kernel<TI, TO, true>), grid, block, 0, st, (const TI*)x, (TO*)y, C, HW);       \
        else                                                                                                         \
            hipLaunchKernelGGL((layout_kernel<TI, TO, false>), grid, block, 0, st, (const TI*)x, (TO*)y, C, HW);      \
    } while (0)
    if (in_dtype == MV_F32 && out_dtype == MV_F32) GO(float, float);
    else if (in_dtype == MV_F32 && out_dtype == MV_BF16) GO(float, bf16_t);
    else if (in_dtype == MV_BF16 && out_dtype == MV_F32) GO(bf16_t, float);
    else GO(bf16_t, bf16_t);
#undef GO
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_nchw_to_nhwc(const void* x, void* y, int N, int C, int H, int W, int in_dtype, int out_dtype,
                    mv_stream_t stream) {
    MV_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0 && N <= 65535, "nchw_to_nhwc: bad args");
    set_kernel_name("nchw_to_nhwc");
    return layout_launch(x, y, N, C, H, W, in_dtype, out_dtype, true, (hipStream_t)stream);
}
int mv_nhwc_to_nchw(const void* x, void* y, int N, int C, int H, int W, int in_dtype, int out_dtype,
                    mv_stream_t stream) {
    MV_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0 && N <= 65535, "nhwc_to_nchw: bad args");
    set_kernel_name("nhwc_to_nchw");
    return layout_launch(x, y, N, C, H, W, in_dtype, out_dtype, false, (hipStream_t)stream);
}

int mv_cast(const void* x, void* y, int64_t n, int in_dtype, int out_dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && y && n > 0, "cast: bad args");
    hipStream_t st = (hipStream_t)stream;
    set_kernel_name("cast");
#define GO(TI, TO) \
    hipLaunchKernelGGL((cast_kernel<TI, TO>), dim3(grid_for(n)), dim3(256), 0, st, (const TI*)x, (TO*)y, (long long)n)
    if (in_dtype == MV_F32 && out_dtype == MV_BF16) GO(float, bf16_t);
    else if (in_dtype == MV_BF16 && out_dtype == MV_F32) GO(bf16_t, float);
    else if (in_dtype == MV_F32) GO(float, float);
    else GO(bf16_t, bf16_t);
#undef GO
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

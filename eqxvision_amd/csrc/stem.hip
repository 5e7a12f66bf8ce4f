// Network-entry convolution for gfx950: raw NCHW image (fp32 or bf16, few channels) -> NHWC bf16
// feature map / token rows, as an implicit GEMM on v_mfma_f32_32x32x16_bf16.
//
// Covers the ResNet stem (7x7/2, reference resnet.py:243-251), AlexNet conv1 (11x11/4, alexnet.py:44),
// ViT PatchEmbed (16x16/16, patch_embed.py:60-62,79-82: output written directly as token rows with the
// position embedding added, vit.py:269) and the Swin patch conv (4x4/4, swin.py:705-711).
//
//   reduction index k = (c, r, s) in OIHW order, so the weight matrix [K][C*R*S] is used as stored;
//   the image operand is gathered on the fly (no im2col buffer): a per-k table in LDS gives the
//   offset of tap k relative to the pixel's top-left input position and its (r, s) for the bounds
//   test; out-of-image taps contribute exact zeros.
//
// Block = 4 waves; each wave owns 32 output pixels x 64 output channels (2 MFMA tiles); the block's
// 64 x Kp weight slab sits in LDS (row pitch chosen so the 32-row ds_read_b128 fragments are
// conflict-free) and is reused by every pixel tile the block walks (grid-stride over pixel tiles).
#include "common.h"

namespace mv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct StemP {
    const void* x;
    const bf16_t* w;
    const float* scale;
    const float* shift;
    const float* pos;
    bf16_t* y;
    int N, C, H, W, K, R, S, Ho, Wo, sh, sw, ph, pw;
    int CRS, Kp, pitch;  // reduction length, padded to 16, LDS row pitch in bytes
    int M, tiles_m, act, tok_stride, tok_offset;
};

template <typename TX> __device__ __forceinline__ float ldx(const TX* p);
template <> __device__ __forceinline__ float ldx<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldx<bf16_t>(const bf16_t* p) { return bf2f(*p); }

template <typename TX>
__global__ __launch_bounds__(256) void stem_kernel(const StemP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [64 rows x pitch] weights | [Kp] int2 tap table
    char* wl = smem;
    int2* ktab = (int2*)(smem + 64 * p.pitch);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * 64;
    const int HW = p.H * p.W;

    // ---- tap table: .x = element offset of tap k from the pixel's (hi0, wi0), .y = r | s<<16 (or -1)
    for (int k = tid; k < p.Kp; k += 256) {
        int2 e;
        if (k < p.CRS) {
            const int c = k / (p.R * p.S);
            const int rs = k - c * p.R * p.S;
            const int r = rs / p.S, s = rs - r * p.S;
            e.x = c * HW + r * p.W + s;
            e.y = r | (s << 16);
        } else {
            e.x = 0;
            e.y = -1;
        }
        ktab[k] = e;
    }
    // ---- weight slab: rows n0..n0+63, zero-padded in n and k
    {
        const int chunks = p.Kp >> 3;  // 16-byte chunks per row
        for (int i = tid; i < 64 * chunks; i += 256) {
            const int row = i / chunks, ch = i - row * chunks;
            const int n = n0 + row;
            uint32_t u[4] = {0, 0, 0, 0};
            if (n < p.K) {
                const bf16_t* src = p.w + (long long)n * p.CRS + ch * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = ch * 8 + e;
                    const uint32_t v = k < p.CRS ? (uint32_t)src[e] : 0u;
                    u[e >> 1] |= v << ((e & 1) * 16);
                }
            }
            *(uint4*)(wl + row * p.pitch + ch * 16) = make_uint4(u[0], u[1], u[2], u[3]);
        }
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const TX* xg = (const TX*)p.x;
    const int nk16 = p.Kp >> 4;

    for (int tile = blockIdx.x; tile < p.tiles_m; tile += gridDim.x) {
        const int m = tile * 128 + wave * 32 + fr;
        const bool mvalid = m < p.M;
        int b = 0, hi0 = 0, wi0 = 0, pix = 0;
        if (mvalid) {
            pix = m % (p.Ho * p.Wo);
            b = m / (p.Ho * p.Wo);
            const int ho = pix / p.Wo, wo = pix - ho * p.Wo;
            hi0 = ho * p.sh - p.ph;
            wi0 = wo * p.sw - p.pw;
        }
        const long long base = (long long)b * p.C * HW + (long long)hi0 * p.W + wi0;

        f32x16 acc[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;

        for (int kk = 0; kk < nk16; ++kk) {
            const int k0 = kk * 16 + fh * 8;
            // image fragment: 8 taps of my pixel
            uint32_t xb[4];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                float v[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int2 t = ktab[k0 + 2 * e2 + q];
                    const int r = t.y & 0xffff, s = t.y >> 16;
                    const bool ok = mvalid && t.y >= 0 && (unsigned)(hi0 + r) < (unsigned)p.H &&
                                    (unsigned)(wi0 + s) < (unsigned)p.W;
                    v[q] = ok ? ldx<TX>(xg + base + t.x) : 0.f;
                }
                xb[e2] = pack_bf2(v[0], v[1]);
            }
            const uint4 bv = make_uint4(xb[0], xb[1], xb[2], xb[3]);
            const uint4 a0 = *(const uint4*)(wl + fr * p.pitch + k0 * 2);
            const uint4 a1 = *(const uint4*)(wl + (32 + fr) * p.pitch + k0 * 2);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0),
                                                             __builtin_bit_cast(bf16x8, bv), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1),
                                                             __builtin_bit_cast(bf16x8, bv), acc[1], 0, 0, 0);
        }

        if (!mvalid) continue;
        long long row = m;
        const float* posr = nullptr;
        if (p.tok_stride > 0) {
            row = (long long)b * p.tok_stride + p.tok_offset + pix;
            if (p.pos) posr = p.pos + (long long)(p.tok_offset + pix) * p.K;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + a * 32 + 8 * g + 4 * fh;
                if (n >= p.K) continue;
                float4 v = make_float4(acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]);
                if (p.scale) {
                    const float4 sc = *(const float4*)(p.scale + n);
                    v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
                }
                if (p.shift) {
                    const float4 sf = *(const float4*)(p.shift + n);
                    v.x += sf.x; v.y += sf.y; v.z += sf.z; v.w += sf.w;
                }
                if (posr) {
                    const float4 pv = *(const float4*)(posr + n);
                    v.x += pv.x; v.y += pv.y; v.z += pv.z; v.w += pv.w;
                }
                if (p.act == MV_ACT_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                } else if (p.act == MV_ACT_GELU_TANH) {
                    v.x = gelu_tanh_f(v.x); v.y = gelu_tanh_f(v.y); v.z = gelu_tanh_f(v.z); v.w = gelu_tanh_f(v.w);
                }
                uint2 u;
                u.x = pack_bf2(v.x, v.y);
                u.y = pack_bf2(v.z, v.w);
                *(uint2*)(p.y + row * p.K + n) = u;
            }
        }
    }
}

int stem_supported(int C, int K, int R, int S, int x_dtype, int out_dtype) {
    const int crs = C * R * S;
    const int kp = (crs + 15) & ~15;
    return out_dtype == MV_BF16 && (x_dtype == MV_F32 || x_dtype == MV_BF16) && K % 4 == 0 && kp <= 1024 &&
           R < 256 && S < 256;
}

int stem_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int C, int H,
                int W, int K, int R, int S, int sh, int sw, int ph, int pw, int act, int x_dtype, int out_dtype,
                int tok_stride, int tok_offset, const float* pos, hipStream_t st) {
    (void)out_dtype;
    StemP p;
    p.x = x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.pos = pos; p.y = (bf16_t*)y;
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.R = R; p.S = S;
    p.Ho = (H + 2 * ph - R) / sh + 1;
    p.Wo = (W + 2 * pw - S) / sw + 1;
    p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
    p.CRS = C * R * S;
    p.Kp = (p.CRS + 15) & ~15;
    p.pitch = ((p.Kp >> 3) | 1) * 16;  // odd number of 16-B slots per row: conflict-free 32-row fragments
    const long long M = (long long)N * p.Ho * p.Wo;
    if (M >= (1LL << 31) - 256 || (long long)N * C * H * W >= (1LL << 31)) {
        set_error("stem conv: tensor too large for 32-bit indexing");
        return MV_E_UNSUPPORTED;
    }
    p.M = (int)M;
    p.tiles_m = (p.M + 127) / 128;
    p.act = act; p.tok_stride = tok_stride; p.tok_offset = tok_offset;
    const size_t smem = (size_t)64 * p.pitch + (size_t)p.Kp * sizeof(int2);
    const int tiles_n = (K + 63) / 64;
    int gx = p.tiles_m;
    const int cap = (256 * 8) / tiles_n > 0 ? (256 * 8) / tiles_n : 1;  // a few resident blocks per CU in total
    if (gx > cap) gx = cap;
    dim3 grid(gx, tiles_n), block(256);
    set_kernel_name(x_dtype == MV_F32 ? "stem_conv_mfma_f32in" : "stem_conv_mfma_bf16in");
    if (x_dtype == MV_F32) {
        if (smem > 48 * 1024)
            MV_HIP(hipFuncSetAttribute((const void*)stem_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem));
        hipLaunchKernelGGL(stem_kernel<float>, grid, block, smem, st, p);
    } else {
        if (smem > 48 * 1024)
            MV_HIP(hipFuncSetAttribute((const void*)stem_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)smem));
        hipLaunchKernelGGL(stem_kernel<bf16_t>, grid, block, smem, st, p);
    }
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // namespace mv

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
#include "mfma_common.h"

namespace mv {

struct StemP {
    const void* x;
    const bf16_t* w;
    const bf16_t* w2;    // optional low halves of split-precision weights (w_true ~ w + w2): second product per k-step
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
    // layout: [64 rows x pitch] weights (| the same for the low halves) | [Kp] int2 tap table
    const bool split = p.w2 != nullptr;                    // block-uniform
    char* wl = smem;
    char* wl2 = smem + 64 * p.pitch;
    int2* ktab = (int2*)(smem + (split ? 128 : 64) * p.pitch);

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
            if (split) {
                uint32_t u2[4] = {0, 0, 0, 0};
                if (n < p.K) {
                    const bf16_t* src = p.w2 + (long long)n * p.CRS + ch * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = ch * 8 + e;
                        const uint32_t v = k < p.CRS ? (uint32_t)src[e] : 0u;
                        u2[e >> 1] |= v << ((e & 1) * 16);
                    }
                }
                *(uint4*)(wl2 + row * p.pitch + ch * 16) = make_uint4(u2[0], u2[1], u2[2], u2[3]);
            }
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
            if (split) {                                   // scalar branch (MFMA ignores EXEC: never a lane mask)
                const uint4 l0 = *(const uint4*)(wl2 + fr * p.pitch + k0 * 2);
                const uint4 l1 = *(const uint4*)(wl2 + (32 + fr) * p.pitch + k0 * 2);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, l0),
                                                                 __builtin_bit_cast(bf16x8, bv), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, l1),
                                                                 __builtin_bit_cast(bf16x8, bv), acc[1], 0, 0, 0);
            }
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
                } else if (p.act > MV_ACT_GELU_TANH) {              // hard_swish / silu stems (MobileNetV3, EfficientNet)
                    v.x = apply_act_rt(v.x, p.act); v.y = apply_act_rt(v.y, p.act); v.z = apply_act_rt(v.z, p.act); v.w = apply_act_rt(v.w, p.act);
                }
                uint2 u;
                u.x = pack_bf2(v.x, v.y);
                u.y = pack_bf2(v.z, v.w);
                *(uint2*)(p.y + row * p.K + n) = u;
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// V1: overlapping windows (ResNet 7x7/2, AlexNet 11x11/4): the block stages the input patch of an
// 8 x 16 output tile in LDS as bf16 (each input element is fetched from HBM/L2 once per tile instead
// of R*S/stride^2 times) and orders the reduction as k = (c, r, s8) with s padded to a multiple of 8
// (zero weights in the padding).  A lane's 8-element MFMA fragment is then 8 CONSECUTIVE patch
// elements of one (c, r) row: four aligned ds_read_b32 instead of 8 gathered global loads.
// The output tile goes through the same LDS transpose as the igemm epilogue: full 128-byte NHWC lines.
// ------------------------------------------------------------------------------------------------
struct StemV1P {
    const void* x;
    const bf16_t* w;
    const float* scale;
    const float* shift;
    bf16_t* y;
    int N, C, H, W, K, R, S, Ho, Wo, sh, sw, ph, pw;
    int SB, nfrag, Kp, wpitch;     // 8-wide s blocks per row, fragments, padded K', weight row pitch (bytes)
    int PH, PWp, patch_elems;      // patch rows, padded row pitch (elements), C*PH*PWp
    int tiles_y, tiles_x, tiles, act;
    int off_patch, off_ftab, off_ep;   // LDS byte offsets
};

// NE > 0: every thread owns NE fixed patch elements and keeps the NEXT tile's values in registers
// (software prefetch issued right after the LDS image of the current tile is written), so the HBM latency of
// the patch gather overlaps the MFMA + epilogue phase instead of stalling each tile twice (~4 us per tile
// measured).  NE = 0: patches too large for that (AlexNet 11x11/4): batched loads inside the tile.
template <typename TX, int NE>
__global__ __launch_bounds__(256) void stem_patch_kernel(const StemV1P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;                                   // [64][wpitch]
    bf16_t* patch = (bf16_t*)(smem + p.off_patch);     // [C][PH][PWp]
    int* ftab = (int*)(smem + p.off_ftab);             // [nfrag_padded] element offset of fragment f in the patch
    constexpr int EPITCH = 64 * 4 + 16;
    char* ep = smem + p.off_ep + (threadIdx.x >> 6) * (32 * EPITCH);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.y * 64;
    const int nk16 = p.Kp >> 4;

    // fragment table + weight slab (k' = (c, r, s8) order, zero padded), once per block
    for (int f = tid; f < 2 * nk16; f += 256) {
        const int ff = f < p.nfrag ? f : p.nfrag - 1;   // padding fragments: any in-bounds address (weights are 0)
        const int sb = ff % p.SB, cr = ff / p.SB;
        const int r = cr % p.R, c = cr / p.R;
        ftab[f] = (c * p.PH + r) * p.PWp + 8 * sb;
    }
    {
        const int chunks = p.Kp >> 3;
        for (int i = tid; i < 64 * chunks; i += 256) {
            const int row = i / chunks, f = i - row * chunks;
            const int n = n0 + row;
            uint32_t u[4] = {0, 0, 0, 0};
            if (n < p.K && f < p.nfrag) {
                const int sb = f % p.SB, cr = f / p.SB;    // cr = c*R + r
                const bf16_t* src = p.w + ((long long)n * p.C * p.R + cr) * p.S + 8 * sb;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t v = (8 * sb + e) < p.S ? (uint32_t)src[e] : 0u;
                    u[e >> 1] |= v << ((e & 1) * 16);
                }
            }
            *(uint4*)(wl + row * p.wpitch + f * 16) = make_uint4(u[0], u[1], u[2], u[3]);
        }
    }

    const int fr = lane & 31, fh = lane >> 5;
    const int py = 2 * wave + (fr >> 4), px = fr & 15;         // my pixel inside the 8 x 16 tile
    const int lbase = (py * p.sh) * p.PWp + px * p.sw;         // patch element offset of my window origin
    const TX* xg = (const TX*)p.x;
    const int HW = p.H * p.W;
    ScaleShift8 ss;
    ss.load(p.scale, p.shift, n0 + (lane & 7) * 8, p.K);

    // prefetch path: per-thread element coordinates are tile independent
    int eoff[NE > 0 ? NE : 1], eyy[NE > 0 ? NE : 1], exx[NE > 0 ? NE : 1];
    float pv[NE > 0 ? NE : 1];
    auto tile_origin = [&](int tile, int& b, int& oy0, int& ox0) {
        const int tx = tile % p.tiles_x;
        const int ty = (tile / p.tiles_x) % p.tiles_y;
        b = tile / (p.tiles_x * p.tiles_y);
        oy0 = ty * 8;
        ox0 = tx * 16;
    };
    auto prefetch = [&](int tile) {
        int b, oy0, ox0;
        tile_origin(tile, b, oy0, ox0);
        const int hi0 = oy0 * p.sh - p.ph, wi0 = ox0 * p.sw - p.pw;
        const TX* xb = xg + (long long)b * p.C * HW;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int hi = hi0 + eyy[j], wi = wi0 + exx[j];
            const bool ok = eoff[j] >= 0 && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const float t = ldx<TX>(xb + (ok ? (long long)eoff[j] + (long long)hi * p.W + wi : 0));
            pv[j] = ok ? t : 0.f;
        }
    };
    if (NE > 0) {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int i = j * 256 + tid;
            const int xx = i % p.PWp;
            const int t2 = i / p.PWp;
            exx[j] = xx;
            eyy[j] = t2 % p.PH;
            eoff[j] = i < p.patch_elems ? (t2 / p.PH) * HW : -1;
        }
        if ((int)blockIdx.x < p.tiles) prefetch(blockIdx.x);
    }

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        int b, oy0, ox0;
        tile_origin(tile, b, oy0, ox0);
        const int hi0 = oy0 * p.sh - p.ph, wi0 = ox0 * p.sw - p.pw;
        __syncthreads();                      // previous tile's MFMAs are done with the patch (and tables are ready)
        if (NE > 0) {
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int i = j * 256 + tid;
                if (i < p.patch_elems) patch[i] = f2bf(pv[j]);
            }
        } else {
            const TX* xb = xg + (long long)b * p.C * HW;
            for (int base = 0; base < p.patch_elems; base += 256 * 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = base + j * 256 + tid;
                    const int xx = i % p.PWp;
                    const int t2 = i / p.PWp;
                    const int yy = t2 % p.PH, c = t2 / p.PH;
                    const int hi = hi0 + yy, wi = wi0 + xx;
                    const bool ok = i < p.patch_elems && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                    const float t = ldx<TX>(xb + (ok ? (long long)c * HW + (long long)hi * p.W + wi : 0));
                    v[j] = ok ? t : 0.f;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int i = base + j * 256 + tid;
                    if (i < p.patch_elems) patch[i] = f2bf(v[j]);
                }
            }
        }
        __syncthreads();
        if (NE > 0 && tile + (int)gridDim.x < p.tiles) prefetch(tile + gridDim.x);   // flies under the MFMAs below

        f32x16 acc[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
        for (int kk = 0; kk < nk16; ++kk) {
            const int f = 2 * kk + fh;
            const uint32_t* src = (const uint32_t*)(patch + lbase + ftab[f]);   // 4-byte aligned (even offsets)
            const uint4 bv = make_uint4(src[0], src[1], src[2], src[3]);
            const uint4 a0 = *(const uint4*)(wl + fr * p.wpitch + f * 16);
            const uint4 a1 = *(const uint4*)(wl + (32 + fr) * p.wpitch + f * 16);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0),
                                                             __builtin_bit_cast(bf16x8, bv), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1),
                                                             __builtin_bit_cast(bf16x8, bv), acc[1], 0, 0, 0);
        }

        // epilogue: scale/shift in the MFMA layout, transpose through the wave's LDS patch, store full lines
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = a * 32 + 8 * g + 4 * fh;
                float4 v = make_float4(acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]);
                *(float4*)(ep + fr * EPITCH + nl * 4) = v;
            }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3), c8 = lane & 7;
            const int oy = oy0 + 2 * wave + (row >> 4), ox = ox0 + (row & 15);
            const int n = n0 + c8 * 8;
            const float4 lo = *(const float4*)(ep + row * EPITCH + c8 * 32);
            const float4 hi = *(const float4*)(ep + row * EPITCH + c8 * 32 + 16);
            if (oy < p.Ho && ox < p.Wo && n < p.K) {
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                ss.apply(v);
                if (p.act == MV_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (p.act == MV_ACT_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
                } else if (p.act > MV_ACT_GELU_TANH) {              // hard_swish / silu stems (MobileNetV3, EfficientNet)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = apply_act_rt(v[e], p.act);
                }
                uint4 u;
                u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
                *(uint4*)(p.y + (((long long)b * p.Ho + oy) * p.Wo + ox) * p.K + n) = u;
            }
        }
    }
}

static bool stem_v1_ok(int C, int K, int R, int S, int sw, int tok_stride) {
    const int SB = (S + 7) / 8;
    const int nfrag = C * R * SB;
    return tok_stride == 0 && (sw % 2) == 0 && K % 8 == 0 && nfrag <= 126 && R <= 16 && S <= 16 && S > 4;
}

static int stem_v1_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int C,
                          int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int act, int x_dtype,
                          hipStream_t st) {
    StemV1P p;
    p.x = x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.y = (bf16_t*)y;
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.R = R; p.S = S;
    p.Ho = (H + 2 * ph - R) / sh + 1;
    p.Wo = (W + 2 * pw - S) / sw + 1;
    p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
    p.SB = (S + 7) / 8;
    p.nfrag = C * R * p.SB;
    p.Kp = ((p.nfrag + 1) / 2) * 16;
    p.wpitch = ((p.Kp >> 3) | 1) * 16;
    p.PH = 7 * sh + R;
    p.PWp = (15 * sw + 8 * p.SB + 1) & ~1;
    p.patch_elems = C * p.PH * p.PWp;
    p.tiles_y = (p.Ho + 7) / 8;
    p.tiles_x = (p.Wo + 15) / 16;
    const long long tiles = (long long)N * p.tiles_y * p.tiles_x;
    if (tiles >= (1LL << 31) || (long long)N * C * H * W >= (1LL << 31)) {
        set_error("stem conv: tensor too large for 32-bit indexing");
        return MV_E_UNSUPPORTED;
    }
    p.tiles = (int)tiles;
    p.act = act;
    p.off_patch = 64 * p.wpitch;
    p.off_ftab = (p.off_patch + p.patch_elems * 2 + 15) & ~15;
    p.off_ep = (p.off_ftab + (p.Kp >> 3) * 4 + 15) & ~15;
    const size_t smem = (size_t)p.off_ep + 4 * 32 * (64 * 4 + 16);
    const int tiles_n = (K + 63) / 64;
    int gx = p.tiles;
    const int cap = (256 * 8) / tiles_n > 0 ? (256 * 8) / tiles_n : 1;
    if (gx > cap) gx = cap;
    dim3 grid(gx, tiles_n), block(256);
    set_kernel_name(x_dtype == MV_F32 ? "stem_patch_mfma_f32in" : "stem_patch_mfma_bf16in");
    const int ne = (p.patch_elems + 255) / 256;
#define GO(TX_, NE_)                                                                                             \
    do {                                                                                                         \
        auto kern = stem_patch_kernel<TX_, NE_>;                                                                 \
        if (smem > 48 * 1024)                                                                                    \
            MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem)); \
        hipLaunchKernelGGL(kern, grid, block, smem, st, p);                                                      \
    } while (0)
    if (x_dtype == MV_F32) {
        if (ne <= 10) GO(float, 10); else GO(float, 0);
    } else {
        if (ne <= 10) GO(bf16_t, 10); else GO(bf16_t, 0);
    }
#undef GO
    MV_LAUNCH_CHECK();
    return MV_OK;
}


// ------------------------------------------------------------------------------------------------
// V2: non-overlapping patches (ViT PatchEmbed k = s = 16 / 8 / 32): a plain GEMM
//     tokens[M = B*P][N = embed] = patches[M][K = C*R*S] . W[N][K]^T  (+ bias + pos_embed)
// whose left operand is never materialised: with S % 8 == 0 an 8-element MFMA fragment is 8
// consecutive pixels of one image row, so every lane fetches its fragments straight from the NCHW
// image (2 x 16-byte loads for fp32 input), converts to bf16 in registers and feeds the B operand.
// The 4 waves of a block own different patches, so the image operand needs no LDS at all; the weight
// tile (shared by the 4 waves) is streamed by LDS-DMA, double buffered, exactly as in igemm.hip.
// Token mode writes row b*tok_stride + tok_offset + p and adds pos_embed (vit.py:269) in the epilogue.
// ------------------------------------------------------------------------------------------------
struct PatchP {
    const void* x;
    const bf16_t* w;
    const float* scale;
    const float* shift;
    const float* pos;
    void* y;               // bf16, or fp32 (ViT tokens entering the fp32 residual stream un-rounded)
    const bf16_t* zero;
    int N, C, H, W, K, R, S, Ho, Wo;
    int CRS, M, tiles_m, tiles_n, act, tok_stride, tok_offset;
};

template <typename TX> struct Frag8;
template <> struct Frag8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
    __device__ __forceinline__ uint4 bf() const {
        return make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
    }
};
template <> struct Frag8<bf16_t> {
    uint4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *(const uint4*)p; }
    __device__ __forceinline__ uint4 bf() const { return v; }
};

template <typename TX, typename OutT>
__global__ __launch_bounds__(256) void patch_embed_kernel(const PatchP p) {
    constexpr int BN = 128, ROWB = 128, TN = 4, WI = BN / 32;
    constexpr int STAGE = BN * ROWB;                       // 16 KB weight tile per stage
    constexpr int EPITCH = 64 * 4 + 16;
    static_assert(4 * 32 * EPITCH <= 3 * STAGE, "epilogue patch must fit");
    __shared__ __attribute__((aligned(16))) char smem[3 * STAGE];   // 2 weight stages (+1 so the epilogue fits)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    int tile_m, tile_n;
    tile_coords(t, p.tiles_m, p.tiles_n, tile_m, tile_n);
    const int m0 = tile_m * 128, n0 = tile_n * BN;
    const int fr = lane & 31, fh = lane >> 5, swz = (fr >> 1) & 7;

    // weight staging (same scheme as igemm.hip: source-side XOR swizzle, lane-linear LDS destination)
    const int srow = lane >> 3;
    const int chunk = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);
    long long woff[WI];
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int n = n0 + 8 * (wave + 4 * j) + srow;
        woff[j] = n < p.K ? (long long)n * p.CRS + chunk * 8 : -1;
    }
    auto stage_w = [&](int buf, int k0) {
        char* ws = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const bf16_t* src = woff[j] >= 0 ? p.w + woff[j] + k0 : p.zero;
            glds16(src, ws + 8 * (wave + 4 * j) * ROWB);
        }
    };

    // my patch: image element offset of its top-left pixel
    const int m = m0 + wave * 32 + fr;
    const int mm = m < p.M ? m : p.M - 1;                  // clamp: loads stay in bounds, the store is masked
    const int P = p.Ho * p.Wo;
    const int pix = mm % P, b = mm / P;
    const int py = pix / p.Wo, px = pix - py * p.Wo;
    const int HW = p.H * p.W;
    const TX* xp = (const TX*)p.x + (long long)b * p.C * HW + (long long)(py * p.R) * p.W + px * p.S;
    const int RS = p.R * p.S;
    auto frag_off = [&](int k) {                           // k = (c, r, s), s multiple of 8
        const int c = k / RS, rem = k - c * RS;
        const int r = rem / p.S, s0 = rem - r * p.S;
        return c * HW + r * p.W + s0;
    };

    ScaleShift8 ss[TN / 2];
#pragma unroll
    for (int c = 0; c < TN / 2; ++c) ss[c].load(p.scale, p.shift, n0 + c * 64 + (lane & 7) * 8, p.K);

    f32x16 acc[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;

    const int nk = p.CRS >> 6;
    Frag8<TX> xf[4], xn[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) xf[kk].load(xp + frag_off(kk * 16 + fh * 8));
    stage_w(0, 0);
    for (int it = 0; it < nk; ++it) {
        const int cur = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* ws = smem + cur * STAGE + fr * ROWB;
        uint4 av[4][TN];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int off = ((2 * kk + fh) ^ swz) << 4;
#pragma unroll
            for (int a = 0; a < TN; ++a) av[kk][a] = *(const uint4*)(ws + a * 32 * ROWB + off);
        }
        if (it + 1 < nk) {
            stage_w(cur ^ 1, (it + 1) * 64);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xn[kk].load(xp + frag_off((it + 1) * 64 + kk * 16 + fh * 8));
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const uint4 bv = xf[kk].bf();
#pragma unroll
            for (int a = 0; a < TN; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[kk][a]),
                                                                 __builtin_bit_cast(bf16x8, bv), acc[a], 0, 0, 0);
        }
        if (it + 1 < nk) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xf[kk] = xn[kk];
        }
    }

    // epilogue (LDS transpose -> full-line stores), + pos_embed rows in token mode
    __syncthreads();
    char* ep = smem + wave * (32 * EPITCH);
#pragma unroll
    for (int c = 0; c < TN / 2; ++c) {
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
            const int a = 2 * c + a2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = a2 * 32 + 8 * g + 4 * fh;
                float4 v = make_float4(acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]);
                *(float4*)(ep + fr * EPITCH + nl * 4) = v;
            }
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3), c8 = lane & 7;
            const int mr = m0 + wave * 32 + row;
            const int n = n0 + c * 64 + c8 * 8;
            const float4 lo = *(const float4*)(ep + row * EPITCH + c8 * 32);
            const float4 hi = *(const float4*)(ep + row * EPITCH + c8 * 32 + 16);
            if (mr < p.M && n < p.K) {
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                ss[c].apply(v);
                long long orow = mr;
                if (p.tok_stride > 0) {
                    const int pb = mr / P, pp = mr - pb * P;
                    orow = (long long)pb * p.tok_stride + p.tok_offset + pp;
                    if (p.pos) {
                        const float* pr = p.pos + (long long)(p.tok_offset + pp) * p.K + n;
                        const float4 p0 = *(const float4*)pr, p1 = *(const float4*)(pr + 4);
                        v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w;
                        v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
                    }
                }
                if (p.act == MV_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (p.act == MV_ACT_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
                } else if (p.act > MV_ACT_GELU_TANH) {              // hard_swish / silu stems (MobileNetV3, EfficientNet)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = apply_act_rt(v[e], p.act);
                }
                Out8<OutT>::st((OutT*)p.y + orow * p.K + n, v);
            }
        }
    }
}

static bool patch_v2_ok(int C, int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int x_dtype) {
    const int crs = C * R * S;
    const int align = x_dtype == MV_F32 ? 4 : 8;
    return sh == R && sw == S && ph == 0 && pw == 0 && S % 8 == 0 && crs % 64 == 0 && K % 8 == 0 && W % align == 0 &&
           H >= R && W >= S;
}

static int patch_v2_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int C,
                           int H, int W, int K, int R, int S, int act, int x_dtype, int out_dtype, int tok_stride, int tok_offset,
                           const float* pos, hipStream_t st) {
    PatchP p;
    p.x = x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.pos = pos; p.y = y;
    p.zero = (const bf16_t*)zero_page(st);
    if (!p.zero) {
        set_error("patch_embed: zero page allocation failed");
        return MV_E_OOM;
    }
    p.N = N; p.C = C; p.H = H; p.W = W; p.K = K; p.R = R; p.S = S;
    p.Ho = (H - R) / R + 1;
    p.Wo = (W - S) / S + 1;
    p.CRS = C * R * S;
    const long long M = (long long)N * p.Ho * p.Wo;
    if (M >= (1LL << 31) - 256 || (long long)N * C * H * W >= (1LL << 31)) {
        set_error("patch_embed: tensor too large for 32-bit indexing");
        return MV_E_UNSUPPORTED;
    }
    p.M = (int)M;
    p.tiles_m = (p.M + 127) / 128;
    p.tiles_n = (K + 127) / 128;
    p.act = act; p.tok_stride = tok_stride; p.tok_offset = tok_offset;
    dim3 grid(p.tiles_m * p.tiles_n), block(256);
    if (out_dtype == MV_F32) {
        set_kernel_name(x_dtype == MV_F32 ? "patch_embed_mfma_f32in_f32out" : "patch_embed_mfma_bf16in_f32out");
        if (x_dtype == MV_F32)
            hipLaunchKernelGGL((patch_embed_kernel<float, float>), grid, block, 0, st, p);
        else
            hipLaunchKernelGGL((patch_embed_kernel<bf16_t, float>), grid, block, 0, st, p);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    set_kernel_name(x_dtype == MV_F32 ? "patch_embed_mfma_f32in" : "patch_embed_mfma_bf16in");
    if (x_dtype == MV_F32)
        hipLaunchKernelGGL((patch_embed_kernel<float, bf16_t>), grid, block, 0, st, p);
    else
        hipLaunchKernelGGL((patch_embed_kernel<bf16_t, bf16_t>), grid, block, 0, st, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int stem_supported(int C, int K, int R, int S, int x_dtype, int out_dtype) {
    const int crs = C * R * S;
    const int kp = (crs + 15) & ~15;
    return out_dtype == MV_BF16 && (x_dtype == MV_F32 || x_dtype == MV_BF16) && K % 4 == 0 && kp <= 1024 &&
           R < 256 && S < 256;
}

// fp32 result (the ViT token rows that START the fp32 residual stream): only the non-overlapping-patch GEMM writes it
int stem_f32out_supported(int C, int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int x_dtype) {
    return (x_dtype == MV_F32 || x_dtype == MV_BF16) && patch_v2_ok(C, H, W, K, R, S, sh, sw, ph, pw, x_dtype) && !get_flag("stem_v0") &&
           !get_flag("no_patch_f32out");
}

int stem_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int C, int H,
                int W, int K, int R, int S, int sh, int sw, int ph, int pw, int act, int x_dtype, int out_dtype,
                int tok_stride, int tok_offset, const float* pos, hipStream_t st, const void* w_lo) {
    if (!w_lo && patch_v2_ok(C, H, W, K, R, S, sh, sw, ph, pw, x_dtype) && !get_flag("stem_v0"))
        return patch_v2_launch(x, w, scale, shift, y, N, C, H, W, K, R, S, act, x_dtype, out_dtype, tok_stride, tok_offset, pos, st);
    if (out_dtype != MV_BF16) {
        set_error("stem conv: fp32 output only from the non-overlapping patch kernel");
        return MV_E_UNSUPPORTED;
    }
    if (!w_lo && stem_v1_ok(C, K, R, S, sw, tok_stride) && !get_flag("stem_v0"))
        return stem_v1_launch(x, w, scale, shift, y, N, C, H, W, K, R, S, sh, sw, ph, pw, act, x_dtype, st);
    StemP p;
    p.x = x; p.w = (const bf16_t*)w; p.w2 = (const bf16_t*)w_lo; p.scale = scale; p.shift = shift; p.pos = pos; p.y = (bf16_t*)y;
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
    const size_t smem = (size_t)(w_lo ? 128 : 64) * p.pitch + (size_t)p.Kp * sizeof(int2);
    const int tiles_n = (K + 63) / 64;
    int gx = p.tiles_m;
    const int cap = (256 * 8) / tiles_n > 0 ? (256 * 8) / tiles_n : 1;  // a few resident blocks per CU in total
    if (gx > cap) gx = cap;
    dim3 grid(gx, tiles_n), block(256);
    set_kernel_name(w_lo ? (x_dtype == MV_F32 ? "stem_conv_mfma_f32in_w2" : "stem_conv_mfma_bf16in_w2")
                         : (x_dtype == MV_F32 ? "stem_conv_mfma_f32in" : "stem_conv_mfma_bf16in"));
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

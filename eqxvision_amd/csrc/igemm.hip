// Implicit-GEMM convolution / Linear for gfx950 (CDNA4), bf16 in, fp32 accumulate, fused epilogue.
//
//   y[m, n] = act( scale[n] * sum_{r,s,c} x[pix(m) + (r,s)][c] * w[n][r][s][c] + shift[n] + res[m, n] )
//
//   m = output pixel (b, ho, wo) of an NHWC feature map (or a row of a [M,K] matrix: Linear = 1x1 conv)
//   n = output channel,  reduction index k = (r, s, c) with c contiguous in BOTH operands.
//
// Design:
//  * no im2col buffer: each lane of the staging wave computes the global address of the 16 bytes
//    (8 channels of one input pixel / one filter tap) it needs and issues ONE
//    `global_load_lds_dwordx4` (LDS-DMA, no VGPR round trip).  Out-of-image taps and rows past the
//    end read a device zero page, so padding costs no branches in the MFMA loop.
//  * k-tile = 64 channels = one 128-byte line per pixel row: 8 lanes fetch a full line.
//  * LDS image: row-major [rows][128 B] with the 16-B slot index XOR-swizzled by (row>>1)&7.  The
//    DMA destination must be lane-linear, so the swizzle is applied to the SOURCE address (which chunk
//    a lane fetches) and again on the fragment read -- `ds_read_b128` is then conflict-free for the
//    32-row x 16-B fragments of v_mfma_f32_32x32x16_bf16.
//  * operands are SWAPPED (A = weights, B = pixels) so that each lane ends up with 4 consecutive
//    output channels of one pixel per accumulator quad: the epilogue (BN scale/shift, residual, ReLU/
//    GELU, bf16 pack) works on float4 and stores 8 bytes per lane straight into the NHWC output.
//  * 256 threads = 4 waves (one per SIMD), double-buffered LDS, one barrier per k-tile: the DMA of
//    tile t+1 is issued before the MFMAs of tile t.
//  * blockIdx -> tile mapping is XCD-aware: the 8 XCDs (private L2s) each take a contiguous chunk of
//    the tile list, n-tiles fastest, so every n-tile of one pixel tile hits the same L2.
#include "common.h"

namespace mv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmP {
    const bf16_t* x;
    const bf16_t* w;
    const float* scale;
    const float* shift;
    const void* residual;
    void* y;
    const bf16_t* zero;
    int N, H, W, C, K, R, S, Ho, Wo, sh, sw, ph, pw, dh, dw;
    int M, tiles_m, tiles_n, act;
};

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    // blocks are dispatched round-robin over the 8 XCDs; give each XCD a contiguous tile range
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename OutT> struct Out4;
template <> struct Out4<bf16_t> {
    __device__ static __forceinline__ float4 ld(const void* p) {
        const uint2 u = *(const uint2*)p;
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    __device__ static __forceinline__ void st(void* p, float4 v) {
        uint2 u;
        u.x = pack_bf2(v.x, v.y);
        u.y = pack_bf2(v.z, v.w);
        *(uint2*)p = u;
    }
};
template <> struct Out4<float> {
    __device__ static __forceinline__ float4 ld(const void* p) { return *(const float4*)p; }
    __device__ static __forceinline__ void st(void* p, float4 v) { *(float4*)p = v; }
};

template <int BM, int BN, int WM, int WN, typename OutT, bool DENSE>
__global__ __launch_bounds__(256) void igemm_bf16_kernel(const IgemmP p) {
    constexpr int ROWB = 128;                 // bytes per LDS row = 64 bf16 of k
    constexpr int XI = BM / 32;               // DMA instructions per thread for the pixel tile
    constexpr int WI = BN / 32;               // ... for the weight tile
    constexpr int TM = BM / WM / 32;          // 32-pixel MFMA tiles per wave
    constexpr int TN = BN / WN / 32;          // 32-channel MFMA tiles per wave
    constexpr int STAGE = (BM + BN) * ROWB;
    static_assert(WM * WN == 4, "4 waves");
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tile_n = t % p.tiles_n, tile_m = t / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---------------- staging constants -------------------------------------------------------
    const int srow = lane >> 3;                                    // row inside the 8-row DMA group
    const int chunk = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);  // which 16-B chunk of the row I fetch
    const int cpt = p.C >> 6;                                      // k-tiles per filter tap
    const int nk = p.R * p.S * cpt;
    const long long wrow_stride = (long long)p.R * p.S * p.C;

    long long xoff[XI];   // DENSE: element offset of the row, -1 = past the end
    int xb[XI], xh[XI], xw[XI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int m = m0 + 8 * (wave + 4 * j) + srow;
        const bool valid = m < p.M;
        if (DENSE) {
            xoff[j] = valid ? (long long)m * p.C + chunk * 8 : -1;
            xb[j] = xh[j] = xw[j] = 0;
        } else {
            const int wo = m % p.Wo;
            const int tt = m / p.Wo;
            const int ho = tt % p.Ho;
            xb[j] = valid ? tt / p.Ho : -1;
            xh[j] = ho * p.sh - p.ph;
            xw[j] = wo * p.sw - p.pw;
            xoff[j] = 0;
        }
    }
    long long woff[WI];
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int n = n0 + 8 * (wave + 4 * j) + srow;
        woff[j] = n < p.K ? (long long)n * wrow_stride + chunk * 8 : -1;
    }

    auto stage = [&](int buf, int r, int s, int c0) {
        char* xs = smem + buf * STAGE;
        char* ws = xs + BM * ROWB;
        const int tapoff = (r * p.S + s) * p.C + c0;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const bf16_t* src = p.zero;
            if (DENSE) {
                if (xoff[j] >= 0) src = p.x + xoff[j] + c0;
            } else {
                const int hi = xh[j] + r * p.dh, wi = xw[j] + s * p.dw;
                if (xb[j] >= 0 && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    src = p.x + (((long long)xb[j] * p.H + hi) * p.W + wi) * p.C + c0 + chunk * 8;
            }
            glds16(src, xs + 8 * (wave + 4 * j) * ROWB);
        }
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const bf16_t* src = woff[j] >= 0 ? p.w + woff[j] + tapoff : p.zero;
            glds16(src, ws + 8 * (wave + 4 * j) * ROWB);
        }
    };

    // ---------------- accumulators ------------------------------------------------------------
    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int wm = wave % WM, wn = wave / WM;
    const int fr = lane & 31, fh = lane >> 5, swz = (fr >> 1) & 7;
    const int xrow0 = wm * (BM / WM), wrow0 = wn * (BN / WN);

    // ---------------- main loop ---------------------------------------------------------------
    int r = 0, s = 0, c0 = 0;   // position of the NEXT tile to stage
    stage(0, r, s, c0);
    for (int it = 0; it < nk; ++it) {
        const int cur = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my DMA pieces of tile `it` have landed
        __syncthreads();                                   // everyone's have; everyone left tile it-1
        // 1) pull the whole k-tile's fragments into registers FIRST: hipcc makes every ds_read issued
        //    after an LDS-DMA wait for vmcnt(0) (it cannot prove the buffers disjoint), which would
        //    serialise the prefetch behind the MFMAs.  Reads-then-DMA keeps the DMA in flight.
        const char* xs = smem + cur * STAGE + xrow0 * ROWB + fr * ROWB;
        const char* ws = smem + cur * STAGE + BM * ROWB + wrow0 * ROWB + fr * ROWB;
        uint4 av[4][TN], bv[4][TM];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int off = ((2 * kk + fh) ^ swz) << 4;
#pragma unroll
            for (int a = 0; a < TN; ++a) av[kk][a] = *(const uint4*)(ws + a * 32 * ROWB + off);
#pragma unroll
            for (int b = 0; b < TM; ++b) bv[kk][b] = *(const uint4*)(xs + b * 32 * ROWB + off);
        }
        // 2) prefetch tile it+1 into the other buffer (asynchronous LDS-DMA)
        if (it + 1 < nk) {
            c0 += 64;
            if (c0 == p.C) {
                c0 = 0;
                if (++s == p.S) {
                    s = 0;
                    ++r;
                }
            }
            stage(cur ^ 1, r, s, c0);
        }
        // 3) 4 k16-steps of MFMA on the register fragments
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av[kk][a]),
                                                                        __builtin_bit_cast(bf16x8, bv[kk][b]),
                                                                        acc[a][b], 0, 0, 0);
    }

    // ---------------- epilogue: D[row = channel][col = pixel] ------------------------------------
    // lane: pixel = fr, channels 8*g + 4*fh + {0..3} for g = 0..3  <-  acc[4*g + {0..3}]
    OutT* y = (OutT*)p.y;
    const OutT* res = (const OutT*)p.residual;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        const int m = m0 + xrow0 + b * 32 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wrow0 + a * 32 + 8 * g + 4 * fh;
                if (n >= p.K) continue;
                float4 v = make_float4(acc[a][b][4 * g + 0], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2],
                                       acc[a][b][4 * g + 3]);
                if (p.scale) {
                    const float4 sc = *(const float4*)(p.scale + n);
                    v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
                }
                if (p.shift) {
                    const float4 sf = *(const float4*)(p.shift + n);
                    v.x += sf.x; v.y += sf.y; v.z += sf.z; v.w += sf.w;
                }
                const long long o = (long long)m * p.K + n;
                if (res) {
                    const float4 rv = Out4<OutT>::ld(res + o);
                    v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                }
                if (p.act == MV_ACT_RELU) {
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                } else if (p.act == MV_ACT_GELU_TANH) {
                    v.x = gelu_tanh_f(v.x); v.y = gelu_tanh_f(v.y); v.z = gelu_tanh_f(v.z); v.w = gelu_tanh_f(v.w);
                }
                Out4<OutT>::st(y + o, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
int igemm_supported(int C, int K, int R, int S, int groups, int in_dtype, int out_dtype) {
    (void)R; (void)S;
    return in_dtype == MV_BF16 && (out_dtype == MV_BF16 || out_dtype == MV_F32) && groups == 1 && C % 64 == 0 &&
           K % 4 == 0;
}

template <int BM, int BN, int WM, int WN>
static int launch_tile(IgemmP& p, bool dense, bool out_f32, hipStream_t st) {
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.K + BN - 1) / BN;
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(256);
#define GO(OT, DN) hipLaunchKernelGGL((igemm_bf16_kernel<BM, BN, WM, WN, OT, DN>), grid, block, 0, st, p)
    if (out_f32) {
        if (dense) GO(float, true); else GO(float, false);
    } else {
        if (dense) GO(bf16_t, true); else GO(bf16_t, false);
    }
#undef GO
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int igemm_launch(const void* x, const void* w, const float* scale, const float* shift, const void* residual, void* y,
                 int N, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw,
                 int act, int in_dtype, int out_dtype, hipStream_t st) {
    (void)in_dtype;
    IgemmP p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.residual = residual; p.y = y;
    p.zero = (const bf16_t*)zero_page(st);
    if (!p.zero) {
        set_error("igemm: zero page allocation failed");
        return MV_E_OOM;
    }
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S;
    p.Ho = (H + 2 * ph - dh * (R - 1) - 1) / sh + 1;
    p.Wo = (W + 2 * pw - dw * (S - 1) - 1) / sw + 1;
    p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw;
    const long long M = (long long)N * p.Ho * p.Wo;
    if (M >= (1LL << 31) - 256) {
        set_error("igemm: M=%lld too large", M);
        return MV_E_UNSUPPORTED;
    }
    p.M = (int)M;
    p.act = act;
    const bool dense = (R == 1 && S == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0);
    const bool out_f32 = out_dtype == MV_F32;
    int tile = get_flag("igemm_tile");
    if (tile == 0) tile = (K <= 64) ? 2 : 1;
    if (tile == 2) {
        set_kernel_name(dense ? "igemm_bf16_128x64_dense" : "igemm_bf16_128x64_conv");
        return launch_tile<128, 64, 4, 1>(p, dense, out_f32, st);
    }
    set_kernel_name(dense ? "igemm_bf16_128x128_dense" : "igemm_bf16_128x128_conv");
    return launch_tile<128, 128, 2, 2>(p, dense, out_f32, st);
}

}  // namespace mv

// Weight-stationary streaming 3x3 convolution for the 64 -> 64 channel layers (ResNet-50 layer1 conv2 at
// 56x56: 3 launches, 59 GFLOP and 205 MB each), gfx950.
//
// With only 64 input channels a k-tile is a whole filter tap, so the tiled igemm kernel ran 9 short
// iterations per block and was dominated by per-block prologue/epilogue (133 us = 0.44 PFLOP/s).  Here, as
// in stream1x1.hip:
//   * the full weight slab W[64][3*3*64] (72 KB) is loaded into LDS once per block and stays;
//   * every wave owns whole 32-pixel tiles and fetches its MFMA B-fragments (16 bytes = 8 channels of one
//     input pixel of one tap) straight from global memory into registers -- the 9-fold tap reuse is served
//     by L1/L2, not by an LDS halo copy; taps are software-pipelined (next tap's 4 fragments in flight while
//     the current tap's 8 MFMAs issue); out-of-image taps read the zero page;
//   * no block barrier in the steady state; epilogue = wave-private LDS transpose + full-line stores.
#include "mfma_common.h"

namespace mv {

struct C3P {
    const bf16_t* x;
    const bf16_t* w;      // KRSC [64][3][3][64]
    const float* scale;
    const float* shift;
    bf16_t* y;
    const bf16_t* zero;
    int N, H, W, M, tiles_m, act;
};

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void conv3x3c64_kernel(const C3P p) {
    constexpr int WPITCH = 9 * 128 + 16;                    // bytes per output-channel row (73 x 16-byte slots: odd)
    constexpr int EPITCH = 64 * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* ep = smem + 64 * WPITCH + wave * (32 * EPITCH);

    {   // weight slab: 64 rows x 72 chunks of 16 bytes
        constexpr int CH = 72;
        for (int base = 0; base < 64 * CH; base += WAVES * 64 * 4) {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = base + j * WAVES * 64 + tid;
                v[j] = *(const uint4*)(p.w + (i < 64 * CH ? (long long)i * 8 : 0));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = base + j * WAVES * 64 + tid;
                const int row = i / CH, ch = i - row * CH;
                if (i < 64 * CH) *(uint4*)(wl + row * WPITCH + ch * 16) = v[j];
            }
        }
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const int gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
    const char* wfrag = wl + fr * WPITCH + fh * 16;             // + a*32*WPITCH + tap*128 + kk*32
    ScaleShift8 ss;
    ss.load(p.scale, p.shift, (lane & 7) * 8, 64);
    const int HW = p.H * p.W;

    // Per-tile lane state: centre-tap pointer of my pixel and the validity of the neighbouring rows / columns.
    struct Pix {
        const bf16_t* px;
        bool c, r0, r2, c0, c2;
    };
    auto locate = [&](int tile) {
        Pix q;
        const int m = tile * 32 + fr;
        q.c = tile < p.tiles_m && m < p.M;
        const int mm = q.c ? m : 0;
        const int b = mm / HW, pix = mm - b * HW;
        const int ho = pix / p.W, wo = pix - ho * p.W;
        q.px = p.x + ((long long)mm * 64) + fh * 8;
        q.r0 = q.c && ho > 0;
        q.r2 = q.c && ho + 1 < p.H;
        q.c0 = wo > 0;
        q.c2 = wo + 1 < p.W;
        return q;
    };
    auto load_tap = [&](uint4* f, const Pix& q, int t) {
        const int r = t / 3, s = t - 3 * r;
        const bool ok = (r == 0 ? q.r0 : (r == 2 ? q.r2 : q.c)) && (s == 0 ? q.c0 : (s == 2 ? q.c2 : true));
        const bf16_t* src = ok ? q.px + ((r - 1) * p.W + (s - 1)) * 64 : p.zero;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) f[kk] = *(const uint4*)(src + kk * 16);
    };

    // Three fragment sets rotate over the taps (tap t lives in set t % 3; 9 taps = 3 rotations, so the
    // rotation lines up across tiles): a set is refilled for the tap three steps ahead -- of this tile or of
    // the wave's NEXT tile -- right after the MFMAs that consumed it, keeping 12 x 1 KB loads in flight per wave.
    // (With one tap of look-ahead the kernel sat in s_waitcnt 82% of the time: SQ_WAIT_ANY, profiles/r01.)
    uint4 fa[4], fb[4], fc[4];
    Pix cur = locate(gw);
    if (gw < p.tiles_m) {
        load_tap(fa, cur, 0);
        load_tap(fb, cur, 1);
        load_tap(fc, cur, 2);
    }
    for (int tile = gw; tile < p.tiles_m; tile += nw) {
        const Pix nxt = locate(tile + nw);
        f32x16 acc[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
        auto mma_tap = [&](const uint4* f, int t) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const uint4 av = *(const uint4*)(wfrag + a * 32 * WPITCH + t * 128 + kk * 32);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                                     __builtin_bit_cast(bf16x8, f[kk]), acc[a], 0, 0, 0);
                }
        };
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            mma_tap(fa, 3 * g);
            if (g < 2) load_tap(fa, cur, 3 * g + 3); else load_tap(fa, nxt, 0);
            mma_tap(fb, 3 * g + 1);
            if (g < 2) load_tap(fb, cur, 3 * g + 4); else load_tap(fb, nxt, 1);
            mma_tap(fc, 3 * g + 2);
            if (g < 2) load_tap(fc, cur, 3 * g + 5); else load_tap(fc, nxt, 2);
        }
        cur = nxt;

        // epilogue
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = a * 32 + 8 * g + 4 * fh;
                *(float4*)(ep + fr * EPITCH + nl * 4) = make_float4(acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2],
                                                                     acc[a][4 * g + 3]);
            }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3), c8 = lane & 7;
            const int mr = tile * 32 + row;
            const float4 lo = *(const float4*)(ep + row * EPITCH + c8 * 32);
            const float4 hi = *(const float4*)(ep + row * EPITCH + c8 * 32 + 16);
            if (mr < p.M) {
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                ss.apply(v);
                if (p.act == MV_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (p.act == MV_ACT_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
                }
                Out8<bf16_t>::st(p.y + (long long)mr * 64 + c8 * 8, v);
            }
        }
    }
}

int conv3x3c64_supported(int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw, int in_dtype,
                         int out_dtype, const void* residual, long long M) {
    return C == 64 && K == 64 && R == 3 && S == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && dh == 1 && dw == 1 &&
           in_dtype == MV_BF16 && out_dtype == MV_BF16 && residual == nullptr && M >= 8192;
}

int conv3x3c64_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int H, int W,
                      int act, hipStream_t st) {
    constexpr int WAVES = 8;
    C3P p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.y = (bf16_t*)y;
    p.zero = (const bf16_t*)zero_page(st);
    if (!p.zero) {
        set_error("conv3x3c64: zero page allocation failed");
        return MV_E_OOM;
    }
    p.N = N; p.H = H; p.W = W;
    p.M = N * H * W;
    p.tiles_m = (p.M + 31) / 32;
    p.act = act;
    const size_t smem = (size_t)64 * (9 * 128 + 16) + (size_t)WAVES * 32 * (64 * 4 + 16);
    int gx = 256;
    const int need = (p.tiles_m + WAVES - 1) / WAVES;
    if (gx > need) gx = need;
    set_kernel_name("conv3x3c64_stream");
    auto kern = conv3x3c64_kernel<WAVES>;
    MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(gx), dim3(WAVES * 64), smem, st, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // namespace mv

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
#include "mfma_common.h"

namespace mv {

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
    int m_off;   // first output row this launch covers (rows m_off .. M-1)
    int xpitch;  // channels per input pixel in memory (== C except in grouped mode)
    int grouped; // > 0: grouped convolution, `grouped` = channels per group (== outputs per group).  C is the per-tap WINDOW (a multiple of
                 // 64) of input channels that the 64 output channels of a tile can touch: it starts at the first channel of the first
                 // group the tile touches, (n0 / grouped) * grouped; pixels are xpitch channels apart; weights [K][R][S][C] hold each
                 // filter at its offset inside its tile's window, zeros elsewhere.
};

// 8 consecutive residual values of one output row, fetched as raw bits early, decoded in the epilogue
template <typename OutT> struct Res8;
template <> struct Res8<bf16_t> {
    uint4 u;
    __device__ __forceinline__ void load(const bf16_t* p) { u = *(const uint4*)p; }
    __device__ __forceinline__ void add_to(float* v) const {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] += __uint_as_float(w[e] << 16);
            v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
        }
    }
};
template <> struct Res8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
    __device__ __forceinline__ void add_to(float* v) const {
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
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
    int tile_m, tile_n;
    tile_coords(t, p.tiles_m, p.tiles_n, tile_m, tile_n);
    const int m0 = p.m_off + tile_m * BM, n0 = tile_n * BN;

    // ---------------- staging constants -------------------------------------------------------
    const int srow = lane >> 3;                                    // row inside the 8-row DMA group
    const int chunk = (lane & 7) ^ ((4 * wave + (lane >> 4)) & 7);  // which 16-B chunk of the row I fetch
    const int cpt = (p.C + 63) >> 6;                               // k-tiles per filter tap
    const int nk = p.R * p.S * cpt;
    const long long wrow_stride = (long long)p.R * p.S * p.C;
    const int goff = p.grouped ? (n0 / p.grouped) * p.grouped : 0;  // grouped: first input channel of the tile's window

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
        // C % 64 != 0: the last k-tile of a tap is zero-filled past C; grouped: so is the part of the window past the last channel
        const bool kin = c0 + chunk * 8 < p.C && goff + c0 + chunk * 8 < p.xpitch;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const bf16_t* src = p.zero;
            if (DENSE) {
                if (xoff[j] >= 0 && kin) src = p.x + xoff[j] + c0;
            } else {
                const int hi = xh[j] + r * p.dh, wi = xw[j] + s * p.dw;
                if (kin && xb[j] >= 0 && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                    src = p.x + (((long long)xb[j] * p.H + hi) * p.W + wi) * p.xpitch + goff + c0 + chunk * 8;
            }
            glds16(src, xs + 8 * (wave + 4 * j) * ROWB);
        }
#pragma unroll
        for (int j = 0; j < WI; ++j) {
            const bf16_t* src = (woff[j] >= 0 && kin) ? p.w + woff[j] + tapoff : p.zero;
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

    // ---------------- residual prefetch ------------------------------------------------------------
    // The epilogue re-reads the tile row-major: in pass q of channel chunk c this lane owns pixel row
    // 8q + lane/8 and channels 64c + 8*(lane%8) .. +7.  Its residual values are fetched NOW, unconditionally
    // (clamped address), so that their HBM latency overlaps the whole main loop instead of forming a chain
    // of 8 dependent round trips at the end.
    const OutT* res = (const OutT*)p.residual;
    ScaleShift8 ss[TN / 2];
#pragma unroll
    for (int c = 0; c < TN / 2; ++c) ss[c].load(p.scale, p.shift, n0 + c * 64 + (lane & 7) * 8, p.K);
    Res8<OutT> rres[TN / 2][4];
    if (res) {
#pragma unroll
        for (int c = 0; c < TN / 2; ++c)
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int m = m0 + xrow0 + pass * 8 + (lane >> 3);
                const int n = n0 + c * 64 + (lane & 7) * 8;
                const bool ok = m < p.M && n < p.K;
                rres[c][pass].load(res + (ok ? (long long)m * p.K + n : 0));
            }
    }

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
            if (c0 >= p.C) {
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

    // ---------------- epilogue -------------------------------------------------------------------
    // MFMA result D[row = channel][col = pixel]: a lane holds pixel `fr` and, per accumulator quad g,
    // channels 8g + 4fh + {0..3}.  Storing that directly touches 32 different 128-byte lines with 16 bytes
    // each per instruction (measured: ~2 TB/s on the memory-bound 1x1 layers).  Instead every wave
    // transposes its 32-pixel x 64-channel chunk through its own LDS patch (fp32, scale/shift already
    // applied) and re-reads it row-major: 8 lanes own one pixel's 64 channels = ONE full 128-byte line of
    // the NHWC output, so the residual read and the store are 16 bytes per lane, fully coalesced.
    constexpr int EPITCH = 64 * 4 + 16;          // bytes per staged pixel row (+16: conflict-free b128 writes)
    static_assert(WN == 1 && TM == 1 && (TN % 2) == 0, "epilogue assumes one 32-pixel tile x BN channels per wave");
    static_assert(4 * 32 * EPITCH <= 2 * STAGE, "epilogue patch must fit the staging buffers");
    __syncthreads();                             // every wave is done reading the last k-tile
    char* ep = smem + wave * (32 * EPITCH);
    OutT* y = (OutT*)p.y;
    const int mrow0 = m0 + xrow0;                // first pixel of this wave
#pragma unroll
    for (int c = 0; c < TN / 2; ++c) {
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {
            const int a = 2 * c + a2;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = a2 * 32 + 8 * g + 4 * fh;
                float4 v = make_float4(acc[a][0][4 * g + 0], acc[a][0][4 * g + 1], acc[a][0][4 * g + 2],
                                       acc[a][0][4 * g + 3]);
                *(float4*)(ep + fr * EPITCH + nl * 4) = v;
            }
        }
        // LDS is in-order per wave: the reads below see the writes above (no barrier needed, the patch
        // is private to the wave)
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3), c8 = lane & 7;
            const int m = mrow0 + row;
            const int n = n0 + c * 64 + c8 * 8;
            const float4 lo = *(const float4*)(ep + row * EPITCH + c8 * 32);
            const float4 hi = *(const float4*)(ep + row * EPITCH + c8 * 32 + 16);
            if (m < p.M && n < p.K) {
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const long long o = (long long)m * p.K + n;
                ss[c].apply(v);
                if (res) rres[c][pass].add_to(v);
                if (p.act == MV_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (p.act == MV_ACT_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
                } else if (p.act != MV_ACT_NONE) {                // hard_swish / hard_sigmoid / sigmoid / silu: this kernel only
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = apply_act_rt(v[e], p.act);
                }
                Out8<OutT>::st(y + o, v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
int igemm_supported(int C, int K, int R, int S, int groups, int in_dtype, int out_dtype) {
    (void)R; (void)S;
    return in_dtype == MV_BF16 && (out_dtype == MV_BF16 || out_dtype == MV_F32) && groups == 1 && C % 64 == 0 &&
           K % 8 == 0;
}

template <int BM, int BN, int WM, int WN>
static int launch_tile(IgemmP& p, bool dense, bool out_f32, hipStream_t st) {
    p.tiles_m = (p.M - p.m_off + BM - 1) / BM;
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

// Per-shape kernel choice: "ov:<M>:<C>:<K>:<R>:<S>:<stride>" flags (tools/tune_tiles.py sets them while it searches:
// greedy, one shape at a time, judged on whole-model ms/step in one process) and the short table of shapes where the
// search beat the rules below by more than its 0.4% threshold.  Choices: 1/2/3 = igemm2 256x64 / 256x128 / 256x256,
// 7 / 8 = 128x128 / 128x64 (this file), 9 = stream1x1, 10 / 11 / 12 = igemm8 256x256 / 128x256 / 256x128;
// 0 = the rules.  Kinds: "ov" conv / linear, "ovh" head-major qkv projection, "ovd" dual-source pointwise (3 = igemm2, 10 = igemm8).
// Round 1 (profiles/r01/*_tile_search.txt): resnet50 B=256 and swin_t B=128 -- nothing above the drift (+0.4% / -0.4%
// in total), rules kept; vit_base B=256, two lanes -- three shapes, +3.5% together:
struct TunedTile { const char* kind; int M, a, b, R, S, sh, choice; };
// Round 2: the three ViT-B rows of round 1 are gone -- igemm8 (choice 10, now the rule for those shapes) beats all of them.
// Round 3 (gpurun_out/r3d/tune_resnet50.log; after bneck_tail changed the mix): resnet50 B=256, two lanes -- two shapes, +2.2% and +1.6%
// (three more "wins" of that log are the rule's own kernel re-measured: drift)
// (inside the laned graph the other lane fills the idle CUs, so the per-FLOP cheaper 256-row tile wins although it leaves
// fewer tiles than CUs); swin_t B=128: nothing above the drift.
static const TunedTile kTuned[] = {
    {"ov", 25088, 1024, 256, 1, 1, 1, 10},    // layer-3 conv1 1x1 1024 -> 256 at 128 images: igemm8 256x256 (98 tiles) over 128x256 (196)
    {"ov", 6272, 512, 512, 3, 3, 1, 12},      // layer-4 conv2 3x3 at 128 images: igemm8 256x128
    {"ov", 21632, 192, 384, 3, 3, 1, 12},     // alexnet conv3 (13 x 13, 192 -> 384) at 128 images: igemm8 256x128 (+0.9%, gpurun_out/r3l/tune_alexnet.log)
};
int tile_override(const char* kind, long long M, int a, int b, int R, int S, int sh) {
    char key[96];
    snprintf(key, sizeof(key), "%s:%lld:%d:%d:%d:%d:%d", kind, M, a, b, R, S, sh);
    const int f = get_flag(key);
    if (f) return f > 0 ? f : 0;                          // a negative flag = "the rules", whatever the table says
    if (get_flag("no_tuned")) return 0;
    for (const TunedTile& t : kTuned)
        if (!strcmp(t.kind, kind) && t.M == M && t.a == a && t.b == b && t.R == R && t.S == S && t.sh == sh) return t.choice;
    return 0;
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
    p.xpitch = C; p.grouped = 0;
    p.Ho = (H + 2 * ph - dh * (R - 1) - 1) / sh + 1;
    p.Wo = (W + 2 * pw - dw * (S - 1) - 1) / sw + 1;
    p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw;
    const long long M = (long long)N * p.Ho * p.Wo;
    if (M >= (1LL << 31) - 256) {
        set_error("igemm: M=%lld too large", M);
        return MV_E_UNSUPPORTED;
    }
    p.M = (int)M;
    p.m_off = 0;
    p.act = act;
    const bool dense = (R == 1 && S == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0);
    const bool out_f32 = out_dtype == MV_F32;
    if (act > MV_ACT_GELU_TANH && dense && !get_flag("no_stream") && !get_flag("igemm_tile") && !get_flag("igemm2_tile") &&
        stream1x1_supported(C, K, in_dtype, out_dtype, M))             // the streaming kernel's epilogue has every activation
        return stream1x1_launch(x, w, scale, shift, residual, y, M, C, K, act, out_dtype, st);
    if (act > MV_ACT_GELU_TANH) {
        // jax.nn.hard_swish / hard_sigmoid / sigmoid / silu (MobileNetV3, EfficientNet): only this file's epilogue implements them;
        // fused here they save the element-wise pass over the layer's output that a faster main loop would not buy back
        if (K <= 64) {
            set_kernel_name(dense ? "igemm_bf16_128x64_dense_act" : "igemm_bf16_128x64_conv_act");
            return launch_tile<128, 64, 4, 1>(p, dense, out_f32, st);
        }
        set_kernel_name(dense ? "igemm_bf16_128x128_dense_act" : "igemm_bf16_128x128_conv_act");
        return launch_tile<128, 128, 4, 1>(p, dense, out_f32, st);
    }
    const bool forced_old = get_flag("igemm_tile") || get_flag("igemm2_tile") || get_flag("no_igemm2");
    if (dense && !get_flag("no_skinny") && !forced_old && get_flag("igemm8") < 2 &&
        skinny_supported(M, C, K, in_dtype, residual))                   // classifier heads: one wave per 32 x 32 tile
        return skinny_launch(x, w, scale, shift, y, M, C, K, act, out_dtype, st);
    const bool ok8 = igemm8_supported(M, C, K, R, S, 2LL * N * H * W * C, 2LL * K * R * S * C);
    if (get_flag("igemm8") >= 2 && get_flag("igemm8") <= 4 && ok8)                                   // forced (tests): 2 = 256x256, 3 = 128x256, 4 = 256x128
        return igemm8_launch(x, w, scale, shift, residual, y, N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, act, out_dtype, 0,
                             get_flag("igemm8") - 1, st);
    const bool plain = !forced_old && !get_flag("no_stream");
    const int ov = plain && sh == sw ? tile_override("ov", M, C, K, R, S, sh) : 0;
    if (ov >= 1 && ov <= 3 && C % 64 == 0 && (long long)R * S * (C / 64) >= 1 && R * S <= 64) {
        igemm2_force_tile(ov);
        const int rc = igemm2_launch(x, w, scale, shift, residual, y, N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, act,
                                     out_dtype, 0, 0, st);
        igemm2_force_tile(0);
        return rc;
    }
    if (ov >= 10 && ov <= 12 && ok8)
        return igemm8_launch(x, w, scale, shift, residual, y, N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, act, out_dtype, 0,
                             ov - 9, st);
    if (ov == 9 && dense && stream1x1_supported(C, K, in_dtype, out_dtype, M))
        return stream1x1_launch(x, w, scale, shift, residual, y, M, C, K, act, out_dtype, st);
    if (ov == 7 || ov == 8) {
        p.m_off = 0;
        if (ov == 8) {
            set_kernel_name(dense ? "igemm_bf16_128x64_dense" : "igemm_bf16_128x64_conv");
            return launch_tile<128, 64, 4, 1>(p, dense, out_f32, st);
        }
        set_kernel_name(dense ? "igemm_bf16_128x128_dense" : "igemm_bf16_128x128_conv");
        return launch_tile<128, 128, 4, 1>(p, dense, out_f32, st);
    }
    if (!get_flag("no_stream") && !get_flag("igemm_tile") && !get_flag("igemm2_tile") &&
        conv3x3c64_supported(C, K, R, S, sh, sw, ph, pw, dh, dw, in_dtype, out_dtype, residual, M))
        return conv3x3c64_v2_launch(x, w, scale, shift, y, N, H, W, act, st);
    if (dense && !get_flag("no_stream") && !get_flag("igemm_tile") && stream1x1_supported(C, K, in_dtype, out_dtype, M))
        return stream1x1_launch(x, w, scale, shift, residual, y, M, C, K, act, out_dtype, st);
    // The ping-pong kernels (igemm8.hip) are THE GEMM core: every layer its rule accepts (enough tiles of one of its three
    // shapes, a reduction of >= 4 k-tiles).  What is left below -- igemm2's 256-row tiles for short reductions, this file's
    // 128 x 128 tile for small / odd shapes -- is the long tail.
    const int t8 = forced_old ? 0 : igemm8_wanted(M, C, K, R, S);
    if (t8 && ok8)
        return igemm8_launch(x, w, scale, shift, residual, y, N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, act, out_dtype, 0, t8,
                             st);
    // deep-pipelined 8-wave kernel (igemm2.hip): real convolutions (taps or stride) and big Linears the rule above passed on
    const long long dense_m = C >= 512 ? 4096 : 32768;
    const bool want2 = igemm2_wanted(M, C, K, R, S) && (!dense || M >= dense_m || out_f32 || get_flag("igemm2_tile"));
    if (!get_flag("no_igemm2") && !get_flag("igemm_tile") && want2)
        return igemm2_launch(x, w, scale, shift, residual, y, N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, act, out_dtype, 0, 0, st);
    p.m_off = 0;
    int tile = get_flag("igemm_tile");
    if (tile == 0) tile = (K <= 64) ? 2 : 1;
    if (tile == 2) {
        set_kernel_name(dense ? "igemm_bf16_128x64_dense" : "igemm_bf16_128x64_conv");
        return launch_tile<128, 64, 4, 1>(p, dense, out_f32, st);
    }
    set_kernel_name(dense ? "igemm_bf16_128x128_dense" : "igemm_bf16_128x128_conv");
    return launch_tile<128, 128, 4, 1>(p, dense, out_f32, st);
}

// Grouped convolution on the matrix cores (ResNeXt conv2, resnet.py:440-471 `groups=32, width_per_group=4|8`; RegNet, regnet.py:49-70,
// group widths 8 ... 264): Cg = C / groups input channels per group and as many output channels.  A tile of 64 output channels
// touches the groups (n0 / Cg) ... ((n0 + 63) / Cg); their input channels form one contiguous WINDOW starting at (n0 / Cg) * Cg.
// The caller expands the filters to [K][R][S][win] (win = the widest window, rounded up to 64): every filter sits at its offset
// inside its tile's window, zeros elsewhere -- sixteen 4-channel groups become ONE 64 -> 64 convolution with a block-diagonal
// weight tile, a 264-channel group a 64 x 320..576 slice.  The k-tile stays a full 128-byte line per pixel, every fragment load is
// aligned (Cg % 8 == 0 or Cg | 64), and the padded MFMA work (win / Cg x) is far cheaper than the scalar kernel (0.9 TFLOP/s).
int igemm_grouped64_supported(int C, int K, int R, int S, int groups, int in_dtype, int out_dtype) {
    if (groups <= 1 || C != K || C % groups != 0) return 0;
    const int cg = C / groups;
    return in_dtype == MV_BF16 && (out_dtype == MV_BF16 || out_dtype == MV_F32) && C % 8 == 0 && (cg % 8 == 0 || 64 % cg == 0) &&
           R * S <= 64;
}

// the widest per-tile input window, rounded up to a multiple of 64 (what the caller's expanded filters must be laid out for)
int igemm_grouped64_window(int C, int groups) {
    const int cg = C / groups;
    int w = 0;
    for (int n0 = 0; n0 < C; n0 += 64) {
        const int last = (n0 + 63 < C ? n0 + 63 : C - 1);
        const int width = (last / cg + 1) * cg - (n0 / cg) * cg;
        if (width > w) w = width;
    }
    return (w + 63) / 64 * 64;
}

int igemm_grouped64_launch(const void* x, const void* w64, const float* scale, const float* shift, const void* residual, void* y,
                           int N, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw, int groups,
                           int act, int out_dtype, hipStream_t st) {
    IgemmP p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w64; p.scale = scale; p.shift = shift; p.residual = residual; p.y = y;
    p.zero = (const bf16_t*)zero_page(st);
    if (!p.zero) {
        set_error("igemm_grouped64: zero page allocation failed");
        return MV_E_OOM;
    }
    p.N = N; p.H = H; p.W = W; p.C = igemm_grouped64_window(C, groups); p.K = K; p.R = R; p.S = S;
    p.xpitch = C; p.grouped = C / groups;
    p.Ho = (H + 2 * ph - dh * (R - 1) - 1) / sh + 1;
    p.Wo = (W + 2 * pw - dw * (S - 1) - 1) / sw + 1;
    p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw;
    p.M = N * p.Ho * p.Wo;
    p.act = act; p.m_off = 0;
    set_kernel_name("igemm_grouped64_bf16_128x64");
    return launch_tile<128, 64, 4, 1>(p, false, out_dtype == MV_F32, st);
}

// Channel counts that are multiples of 8 but not of 64 (MobileNet-style widths: 16, 24, 32, 96, 144, 160 ...): the 128 x 128 /
// 128 x 64 kernel of this file with the last k-tile of every tap zero-filled past C.
int igemm_oddc_supported(int C, int K, int groups, int in_dtype, int out_dtype) {
    return in_dtype == MV_BF16 && (out_dtype == MV_BF16 || out_dtype == MV_F32) && groups == 1 && C % 8 == 0 && C % 64 != 0 &&
           K % 8 == 0;
}

int igemm_oddc_launch(const void* x, const void* w, const float* scale, const float* shift, const void* residual, void* y, int N,
                      int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw, int act,
                      int out_dtype, hipStream_t st) {
    IgemmP p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.residual = residual; p.y = y;
    p.zero = (const bf16_t*)zero_page(st);
    if (!p.zero) {
        set_error("igemm_oddc: zero page allocation failed");
        return MV_E_OOM;
    }
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S;
    p.xpitch = C; p.grouped = 0;
    p.Ho = (H + 2 * ph - dh * (R - 1) - 1) / sh + 1;
    p.Wo = (W + 2 * pw - dw * (S - 1) - 1) / sw + 1;
    p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw;
    p.M = N * p.Ho * p.Wo;
    p.act = act; p.m_off = 0;
    const bool dense = (R == 1 && S == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0);
    if (K <= 64) {
        set_kernel_name(dense ? "igemm_bf16_128x64_dense_oddc" : "igemm_bf16_128x64_conv_oddc");
        return launch_tile<128, 64, 4, 1>(p, dense, out_dtype == MV_F32, st);
    }
    set_kernel_name(dense ? "igemm_bf16_128x128_dense_oddc" : "igemm_bf16_128x128_conv_oddc");
    return launch_tile<128, 128, 4, 1>(p, dense, out_dtype == MV_F32, st);
}

}  // namespace mv

// A whole ResNet bottleneck of the 56 x 56 stage in ONE launch, row-strip tiled (gfx950):
//
//     t1 = relu(bn1(conv1_1x1(x)))     t2 = relu(bn2(conv2_3x3(t1)))     y = relu(bn3(conv3_1x1(t2)) + identity)
//
// (reference resnet.py:144-162), identity = x for the two identity blocks of layer 1 (CIN = 256) and bn_d(conv_d_1x1(x)) for
// its first block (CIN = 64, `DUAL`: conv3 and the downsample convolution are one reduction over [t2 | x] with the BatchNorm
// scales folded into the bf16 weight rows, resnet.py:295-303).  The 64-channel intermediates t1 and t2 never reach HBM and the
// block input is read ONCE: it is both the operand of conv1 and the identity.  Un-fused (conv3x3c64 + chain1x1) a block moves
// 617 MB per 128 images, this kernel 461 MB (x with its halo rows + y).
//
// One 512-thread workgroup owns a strip of 8 rows x 56 columns of one image = 448 positions = 14 MFMA column blocks of 32:
//   * waves 0-6 own two blocks (64 consecutive positions) each, through all three products ("position-major"):
//       A  conv1: B = the wave's x rows, loaded straight from HBM in fragment layout (a lane = one position x 8 channels per
//          k16-step) and KEPT in registers (128 VGPRs at CIN = 256) until the identity is added; A = W1 fragments streamed
//          from L2 (host prepared fragment order, rolling prefetch).  BN + ReLU, bf16 -> the strip's t1 map in LDS: 10 rows x
//          58 columns of 144-byte slots (64 channels + 16 bytes of padding: conflict-free ds_read_b128 at consecutive slots),
//          zero border columns / out-of-image rows.
//       B  conv2: B = t1 slots (ONE base address per block, every (tap, k-step) is an instruction offset), A = W2 fragments
//          from LDS (72 KB, copied once per workgroup); no barrier inside.  BN + ReLU, bf16 -> registers.
//       C  conv3: t2 goes through a wave-private LDS patch into B-fragment layout; per 32-channel tile 4 (DUAL: 8, the second
//          four on the x fragments still in registers) MFMAs with W3 fragments from LDS; fp32 patch transpose, scale / shift,
//          + identity (the x registers written row-major into a second patch), ReLU, 16-byte stores.
//   * wave 7 computes t1 for the two halo rows (the row above and the row below the strip: conv1 is recomputed there, +25 % of
//     conv1 = +6 % of the block's FLOPs) and is idle afterwards.
// Rounding points are those of the un-fused launches (t1, t2 bf16; fp32 accumulate, fp32 epilogue, one rounding of y).
#include "mfma_common.h"

namespace mv {

namespace {

struct StripP {
    const bf16_t* x;      // [B][56][56][CIN]
    const bf16_t* w1f;    // [2][CIN/16][64 lanes][8]
    const float* s1;      // [64]
    const float* h1;
    const bf16_t* w2f;    // [2][9 taps][4][64 lanes][8]
    const float* s2;
    const float* h2;
    const bf16_t* w3f;    // [8][KC3][64 lanes][8]; KC3 = 4, DUAL: 8 = [scale3 * W3 | scale_d * W_d]
    const float* s3;      // [256] (DUAL: ones)
    const float* h3;      // [256] (DUAL: shift3 + shift_d)
    bf16_t* y;            // [B][56][56][256]
    long long* prof;      // debug build (MV_I8_PROF) only: per-wave wall-clock stamps at the phase boundaries
    int skew;             // first-round workgroups start (block / 8 % 4) x skew x 10 ns late: de-phases the CUs (see below)
};

constexpr int HW = 56, RS = 8, NSTRIP = HW / RS;
constexpr int TP = HW + 2;                       // slots per t1 map row
constexpr int SLOTB = 144;                       // bytes per slot
constexpr int T1B = (RS + 2) * TP * SLOTB;       // 83 520
constexpr int W2B = 2 * 36 * 1024;               // 73 728
constexpr int LDS_AB = T1B + W2B;                // 157 248
constexpr int PATCHB = 32 * SLOTB;               // 4 608: [32 rows][64 bf16 | 32 fp32] + pad
constexpr int RESP = 80;                         // residual patch row: 32 bf16 + pad
constexpr int WAREA = 2 * PATCHB + 32 * RESP;    // 11 776

template <int CIN, bool DUAL>
__global__ __launch_bounds__(512) void bneck_strip_kernel(const StripP p) {
    constexpr int KC1 = CIN / 16;                // k16-steps of conv1
    constexpr int KC3 = DUAL ? 8 : 4;
    constexpr int W3B = 8 * KC3 * 1024;
    constexpr int D1 = 4;                        // W1 fragments in flight (k-steps)
    static_assert(KC1 % D1 == 0, "prefetch ring");
    static_assert(W3B + 7 * WAREA <= LDS_AB, "phase C layout fits the phase A/B allocation");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const t1m = smem;
    char* const w2l = smem + T1B;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int img = blockIdx.x / NSTRIP, strip = blockIdx.x - img * NSTRIP;
    const int r0 = strip * RS;
    const bool top_in = strip > 0, bot_in = strip < NSTRIP - 1;
    const size_t pix0 = (size_t)img * (HW * HW) + (size_t)r0 * HW;       // first own pixel
#ifdef MV_I8_PROF
    long long st_w[8];
#define MV_BS_STAMP(i) do { if (p.prof) st_w[i] = wall_clock64(); } while (0)
#else
#define MV_BS_STAMP(i) do {} while (0)
#endif
#ifdef MV_I8_PROF                                    // the stagger experiment (debug build): phase A stays ~21 us per workgroup with it --
    if (p.skew > 0 && blockIdx.x < 256) {            // a CU's memory pipe, not chip-wide contention, is what bounds it
        const int q = (blockIdx.x >> 3) & 3;
        if (q) {
            const long long t_end = wall_clock64() + (long long)q * p.skew;
            while (wall_clock64() < t_end) __builtin_amdgcn_s_sleep(32);
        }
    }
#endif
    MV_BS_STAMP(0);

    // ---------------- zero border of the t1 map ---------------------------------------------------------------------------
    if (tid < 2 * (RS + 2) * 9) {                                         // columns 0 and 57 of every row: 20 slots x 9 chunks
        const int s = tid / 9, c = tid - s * 9;
        const int slot = (s >> 1) * TP + (s & 1) * (TP - 1);
        *(uint4*)(t1m + slot * SLOTB + c * 16) = make_uint4(0, 0, 0, 0);
    }
    if (!top_in)
        for (int i = tid; i < TP * 9; i += 512) *(uint4*)(t1m + i * 16) = make_uint4(0, 0, 0, 0);
    if (!bot_in)
        for (int i = tid; i < TP * 9; i += 512) *(uint4*)(t1m + (RS + 1) * TP * SLOTB + i * 16) = make_uint4(0, 0, 0, 0);

    // ---------------- W2 fragments -> LDS by LDS-DMA (lane-linear copy; in flight under phase A) --------------------------
    constexpr int NW2 = W2B / 16;                                          // 4608 chunks of 16 bytes = 9 per thread
    static_assert(NW2 % 512 == 0, "whole DMA instructions");
#pragma unroll
    for (int i = 0; i < NW2 / 512; ++i)
        glds16(p.w2f + (size_t)(tid + 512 * i) * 8, w2l + (wave * 64 + 512 * i) * 16);

    // ---------------- phase A: conv1 on two blocks of 32 positions ----------------------------------------------------------
    // q0 / q1: first position of the two blocks, counted from the strip's halo row (q = 56 (row + 1) + col, row -1 .. 8);
    // n0 / n1: valid positions in them (a halo row ends after 24 positions of its second block)
    uint4 xk[2][KC1];
    auto conv1_pair = [&](int q0, int q1, int n0, int n1) {
        const uint4* ap = (const uint4*)p.w1f + lane;
        uint4 a1[D1][2];
#pragma unroll
        for (int d = 0; d < D1; ++d) {
            a1[d][0] = ap[(0 * KC1 + d) * 64];
            a1[d][1] = ap[(1 * KC1 + d) * 64];
        }
        const int qa = q0 + (fr < n0 ? fr : n0 - 1), qb = q1 + (fr < n1 ? fr : n1 - 1);      // clamped: never stored
        const bf16_t* xa = p.x + ((pix0 - HW) + qa) * CIN + 8 * fh;
        const bf16_t* xb = p.x + ((pix0 - HW) + qb) * CIN + 8 * fh;
#pragma unroll
        for (int j = 0; j < KC1; ++j) {
            xk[0][j] = *(const uint4*)(xa + 16 * j);
            xk[1][j] = *(const uint4*)(xb + 16 * j);
        }
        f32x16 acc[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][a][e] = 0.f;
#pragma unroll
        for (int j = 0; j < KC1; ++j) {
            const bf16x8 af0 = __builtin_bit_cast(bf16x8, a1[j % D1][0]);
            const bf16x8 af1 = __builtin_bit_cast(bf16x8, a1[j % D1][1]);
            if (j + D1 < KC1) {
                a1[j % D1][0] = ap[(0 * KC1 + j + D1) * 64];
                a1[j % D1][1] = ap[(1 * KC1 + j + D1) * 64];
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const bf16x8 bf = __builtin_bit_cast(bf16x8, xk[b][j]);
                acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af0, bf, acc[b][0], 0, 0, 0);
                acc[b][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1, bf, acc[b][1], 0, 0, 0);
            }
        }
        // BN + ReLU -> bf16 -> t1 map (slot = 58 row + col + 1)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int q = (b ? q1 : q0) + fr;
            const bool ok = fr < (b ? n1 : n0);
            const int row = q / HW, col = q - row * HW;
            char* dst = t1m + (row * TP + col + 1) * SLOTB + 8 * fh;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = 32 * a + 8 * g + 4 * fh;
                    const float4 sc = *(const float4*)(p.s1 + ch), sh = *(const float4*)(p.h1 + ch);
                    uint2 u;
                    u.x = pack_bf2(fmaxf(fmaf(acc[b][a][4 * g + 0], sc.x, sh.x), 0.f), fmaxf(fmaf(acc[b][a][4 * g + 1], sc.y, sh.y), 0.f));
                    u.y = pack_bf2(fmaxf(fmaf(acc[b][a][4 * g + 2], sc.z, sh.z), 0.f), fmaxf(fmaf(acc[b][a][4 * g + 3], sc.w, sh.w), 0.f));
                    if (ok) *(uint2*)(dst + (32 * a + 8 * g) * 2) = u;
                }
        }
    };

    MV_BS_STAMP(1);
    // own positions of this wave: p = 64 wave + 32 b + fr (wave < 7)
    const int pw = 64 * wave;
    if (wave < 7) {
        conv1_pair(HW + pw, HW + pw + 32, 32, 32);
    } else {
        if (top_in) conv1_pair(0, 32, 32, HW - 32);
        if (bot_in) conv1_pair((RS + 1) * HW, (RS + 1) * HW + 32, 32, HW - 32);
    }
    MV_BS_STAMP(2);
    __syncthreads();
    MV_BS_STAMP(3);

    // ---------------- phase B: conv2 3x3 over the t1 map --------------------------------------------------------------------
    uint2 t2p[2][8];                                      // t2 of the wave's two blocks, bf16, accumulator layout
    if (wave < 7) {
        f32x16 acc[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][a][e] = 0.f;
        const char* bb[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int pp = pw + 32 * b + fr;
            const int row = pp / HW, col = pp - row * HW;
            bb[b] = t1m + (row * TP + col) * SLOTB + 16 * fh;           // slot of filter tap (0, 0) = the upper left neighbour
        }
        const char* wl = w2l + lane * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int boff = ((tap / 3) * TP + (tap % 3)) * SLOTB + 32 * j;
                const bf16x8 af0 = *(const bf16x8*)(wl + ((0 * 36 + tap * 4 + j) << 10));
                const bf16x8 af1 = *(const bf16x8*)(wl + ((1 * 36 + tap * 4 + j) << 10));
                const bf16x8 b0 = *(const bf16x8*)(bb[0] + boff);
                const bf16x8 b1 = *(const bf16x8*)(bb[1] + boff);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1, b0, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af0, b1, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af1, b1, acc[1][1], 0, 0, 0);
            }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = 32 * a + 8 * g + 4 * fh;
                const float4 sc = *(const float4*)(p.s2 + ch), sh = *(const float4*)(p.h2 + ch);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    uint2 u;
                    u.x = pack_bf2(fmaxf(fmaf(acc[b][a][4 * g + 0], sc.x, sh.x), 0.f), fmaxf(fmaf(acc[b][a][4 * g + 1], sc.y, sh.y), 0.f));
                    u.y = pack_bf2(fmaxf(fmaf(acc[b][a][4 * g + 2], sc.z, sh.z), 0.f), fmaxf(fmaf(acc[b][a][4 * g + 3], sc.w, sh.w), 0.f));
                    t2p[b][4 * a + g] = u;
                }
            }
    }
    MV_BS_STAMP(4);
    __syncthreads();                                      // nobody reads the t1 map or the W2 fragments any more

    // ---------------- phase C: conv3 1x1 + BN + identity + ReLU -------------------------------------------------------------
    {
        constexpr int NW3 = W3B / 16;
        static_assert(NW3 % 512 == 0, "whole DMA instructions");
#pragma unroll
        for (int i = 0; i < NW3 / 512; ++i)
            glds16(p.w3f + (size_t)(tid + 512 * i) * 8, smem + (wave * 64 + 512 * i) * 16);
    }
    __syncthreads();
    MV_BS_STAMP(5);
#ifdef MV_I8_PROF
    if (wave >= 7 && p.prof && lane == 0) {
        long long* o = p.prof + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 6; ++i) o[i] = st_w[i];
        o[6] = o[7] = st_w[5];
    }
#endif
    if (wave >= 7) return;

    char* const tp = smem + W3B + wave * WAREA;           // t2 patch  [32][64 bf16]
    char* const fp = tp + PATCHB;                         // fp32 patch [32][32 fp32]
    char* const rp = fp + PATCHB;                         // identity patch [32][32 bf16]
    const char* const w3l = smem + lane * 16;
    const int er = lane >> 2, ec = lane & 3;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        // t2 of block b: accumulator layout -> B-fragment layout through the patch
#pragma unroll
        for (int i = 0; i < 8; ++i) *(uint2*)(tp + fr * SLOTB + (32 * (i >> 2) + 8 * (i & 3) + 4 * fh) * 2) = t2p[b][i];
        wave_lds_fence();
        bf16x8 tb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) tb[j] = *(const bf16x8*)(tp + fr * SLOTB + (2 * j + fh) * 16);
        wave_lds_fence();
        bf16_t* const yb = p.y + (pix0 + pw + 32 * b) * 256;
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(w3l + ((a * KC3 + j) << 10)), tb[j], acc, 0, 0, 0);
            if constexpr (DUAL) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(w3l + ((a * KC3 + 4 + j) << 10)),
                                                                  __builtin_bit_cast(bf16x8, xk[b][j]), acc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(fp + fr * SLOTB + (8 * g + 4 * fh) * 4) = make_float4(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            if constexpr (!DUAL) {                         // identity: x channels 32 a .. 32 a + 31 of the block's positions, row-major
                *(uint4*)(rp + fr * RESP + 16 * fh) = xk[b][2 * a];
                *(uint4*)(rp + fr * RESP + 32 + 16 * fh) = xk[b][2 * a + 1];
            }
            wave_lds_fence();
            const int n = 32 * a + 8 * ec;
            const float4 sca = *(const float4*)(p.s3 + n), scb = *(const float4*)(p.s3 + n + 4);
            const float4 sha = *(const float4*)(p.h3 + n), shb = *(const float4*)(p.h3 + n + 4);
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int row = 16 * ps + er;
                const float4 lo = *(const float4*)(fp + row * SLOTB + ec * 32);
                const float4 hi = *(const float4*)(fp + row * SLOTB + ec * 32 + 16);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                v[0] = fmaf(v[0], sca.x, sha.x); v[1] = fmaf(v[1], sca.y, sha.y);
                v[2] = fmaf(v[2], sca.z, sha.z); v[3] = fmaf(v[3], sca.w, sha.w);
                v[4] = fmaf(v[4], scb.x, shb.x); v[5] = fmaf(v[5], scb.y, shb.y);
                v[6] = fmaf(v[6], scb.z, shb.z); v[7] = fmaf(v[7], scb.w, shb.w);
                if constexpr (!DUAL) {
                    const uint4 rr = *(const uint4*)(rp + row * RESP + ec * 16);
                    const uint32_t w[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[2 * e] += __uint_as_float(w[e] << 16);
                        v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                Out8<bf16_t>::st(yb + (size_t)row * 256 + n, v);
            }
            wave_lds_fence();
        }
        MV_BS_STAMP(6 + b);
    }
#ifdef MV_I8_PROF
    if (p.prof && lane == 0) {
        long long* o = p.prof + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 8; ++i) o[i] = st_w[i];
    }
#endif
#undef MV_BS_STAMP
}

}  // namespace

}  // namespace mv

extern "C" {

int mv_bottleneck_strip_supported(int H, int W, int cin, int width, int cout, int dual, int dtype) {
    return dtype == MV_BF16 && H == 56 && W == 56 && width == 64 && cout == 256 && ((!dual && cin == 256) || (dual && cin == 64));
}

int mv_bottleneck_strip_fwd(const void* x, const void* w1f, const float* scale1, const float* shift1, const void* w2f,
                            const float* scale2, const float* shift2, const void* w3f, const float* scale3, const float* shift3,
                            void* y, int B, int H, int W, int cin, int width, int cout, int dual, int dtype, mv_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    using namespace mv;
    MV_CHECK_ARG(x && w1f && scale1 && shift1 && w2f && scale2 && shift2 && w3f && scale3 && shift3 && y,
                 "mv_bottleneck_strip_fwd: null argument");
    MV_CHECK_ARG(B > 0 && (long long)B * 7 < (1LL << 31), "mv_bottleneck_strip_fwd: B = %d", B);
    if (!mv_bottleneck_strip_supported(H, W, cin, width, cout, dual, dtype)) {
        set_error("mv_bottleneck_strip_fwd: unsupported configuration %dx%d %d -> %d -> %d dual %d (ask mv_bottleneck_strip_supported first)",
                  H, W, cin, width, cout, dual);
        return MV_E_UNSUPPORTED;
    }
    StripP p;
    p.x = (const bf16_t*)x; p.w1f = (const bf16_t*)w1f; p.s1 = scale1; p.h1 = shift1;
    p.w2f = (const bf16_t*)w2f; p.s2 = scale2; p.h2 = shift2;
    p.w3f = (const bf16_t*)w3f; p.s3 = scale3; p.h3 = shift3; p.y = (bf16_t*)y;
    p.prof = nullptr;
    p.skew = 0;
#ifdef MV_I8_PROF
    p.skew = get_flag("strip_skew");
    if (get_flag("bneck_prof"))
        p.prof = (long long*)(((unsigned long long)(unsigned)get_flag("prof_hi") << 32) | (unsigned)get_flag("prof_lo"));
#endif
    const dim3 grid((unsigned)(B * NSTRIP)), block(512);
    static LdsAttrSite attr[2];
    if (dual) {
        auto kern = bneck_strip_kernel<64, true>;
        MV_HIP(attr[1].ensure((const void*)kern, LDS_AB));
        set_kernel_name("bneck_strip_dual_bf16_56x56_64_64_256");
        hipLaunchKernelGGL(kern, grid, block, LDS_AB, stream, p);
    } else {
        auto kern = bneck_strip_kernel<256, false>;
        MV_HIP(attr[0].ensure((const void*)kern, LDS_AB));
        set_kernel_name("bneck_strip_bf16_56x56_256_64_256");
        hipLaunchKernelGGL(kern, grid, block, LDS_AB, stream, p);
    }
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

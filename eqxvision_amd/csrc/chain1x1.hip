// Two pointwise layers in one streaming kernel (ResNet-50 layer1, 56x56): the last convolution of a bottleneck and
// the first convolution of the NEXT one (reference resnet.py:144-162, blocks chained by nn.Sequential :330-333),
//
//   y [m, 0:256] = relu( scale3 * (W3  t2[m, 0:64]) + shift3 + res[m, 0:256] )        -> HBM (the next block's identity)
//   t1[m, 0:64]  = relu( scale1 * (W1n y [m, 0:256]) + shift1 )                        -> HBM (the next block's conv2 input)
//
// Each of the two layers alone runs at the HBM roofline (stream1x1.hip, 5.3 TB/s); what is left to save is traffic:
// un-fused the 256-channel map y is written once and read twice (as the next conv1's input and as the next residual).
// Here the second GEMM consumes y while a wave still holds it: -411 MB per block boundary at batch 256.
//
// Structure = stream1x1.hip with both 128-channel slabs of W3 resident (weights in LDS, every wave free-runs over
// 32-pixel tiles, x fragments straight from HBM into registers, no block barrier in the steady state), plus:
//   * after a 64-channel chunk of y has been finished (scale / shift / residual / ReLU, rounded to bf16, stored), the
//     same bf16 values are written back into the wave's LDS patch as [32 pixels][64 channels] -- exactly the B operand
//     layout of the second GEMM -- and 8 MFMAs accumulate W1n[:, chunk] * y_chunk into two more accumulators;
//   * after the fourth chunk the 32 x 64 result goes through the usual patch transpose and is stored as whole lines.
// The second GEMM therefore sees y exactly as the next layer would read it from HBM (bf16-rounded): bit-identical to
// the un-fused pair.
#include "mfma_common.h"

namespace mv {

struct ChainP {
    const bf16_t* x;        // t2 [M][64]
    const bf16_t* x2;       // DUAL: the block input [M][64] (second K-source: the downsample branch)
    const bf16_t* w3;       // [256][64]; DUAL: [256][128] = [scale3 * W3 | scale_d * W_d]
    const float* scale3;
    const float* shift3;
    const bf16_t* residual; // [M][256]
    bf16_t* y;              // [M][256]
    const bf16_t* w1;       // [64][256]
    const float* scale1;
    const float* shift1;
    bf16_t* t1;             // [M][64]
    int M, tiles_m;
    int subH, subW;         // SUB: the map is [.., subH, subW] and y is its stride-2 sub-sampling [.., subH / 2, subW / 2, 256]
};

// N2 = 64: the next block of the same stage (8 waves).  N2 = 128: the first block of the next stage, whose conv1 is
// still stride 1 on the 56x56 map (ResNet v1.5 strides the 3x3); its weights take 66 KB, so 6 waves' patches fit.
// DUAL: the first block of layer1, whose identity is itself a 1x1 convolution of the block input (downsample = conv + BN,
// resnet.py:295-303).  conv3 and the downsample conv write the same output, so they are ONE GEMM over the concatenated
// reduction [t2 | x] with the two BatchNorm scales folded into the bf16 weight rows (shifts summed): the 256-channel
// identity map is neither written nor read (-411 MB at batch 256) and its launch disappears.
// SUB (round 6): the ONLY consumer of y besides the fused conv1 is a stride-2 pointwise convolution (the downsample branch of the next
// stage's first block, resnet.py:295-303; the 3x3 of ResNet v1.5 strides on t1, not on y): only the pixels with even (h, w) are
// stored, compactly -- a quarter of the 256-channel map's bytes -- and the strided consumer reads that tensor with stride 1.
// y == NULL (DUAL, round 6): nobody reads this block's output map (the next block recomputes it, chain_rc.hip): no store.
template <int N2, int WAVES, bool DUAL, bool SUB = false>
__global__ __launch_bounds__(WAVES * 64) void chain1x1_kernel(const ChainP p) {
    constexpr int C = DUAL ? 128 : 64, K = 256;
    constexpr int T2 = N2 / 32;                                 // 32-channel tiles of the second GEMM
    constexpr int KC = C / 16;                                  // 4 k16-steps of the first GEMM
    constexpr int W3P = C * 2 + 16;                             // 144: odd number of 16-byte slots
    constexpr int W1P = K * 2 + 16;                             // 528
    constexpr int EPITCH = 64 * 4 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w3l = smem;                                           // [256][W3P]
    char* w1l = w3l + K * W3P;                                  // [64][W1P]
    float* sct = (float*)(w1l + N2 * W1P);                      // scale3[256], shift3[256], scale1[64], shift1[64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* ep = (char*)(sct + 2 * K + 2 * N2) + wave * (32 * EPITCH);

    // both weight matrices -> LDS, 16-byte chunks, several loads in flight per thread (one load per loop trip made the
    // prologue a chain of dependent L2 round trips)
    {
        constexpr int NT = WAVES * 64, N3 = K * (C / 8), N1 = N2 * (K / 8), U = 4;
        for (int base = 0; base < N3; base += U * NT) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * NT + tid;
                const int ic = i < N3 ? i : N3 - 1;
                v[u] = *(const uint4*)(p.w3 + (long long)ic * 8);           // rows are C = 8 * (C/8) elements: contiguous
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * NT + tid;
                if (i < N3) *(uint4*)(w3l + (i / (C / 8)) * W3P + (i % (C / 8)) * 16) = v[u];
            }
        }
        for (int base = 0; base < N1; base += U * NT) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * NT + tid;
                const int ic = i < N1 ? i : N1 - 1;
                v[u] = *(const uint4*)(p.w1 + (long long)(ic >> 5) * K + (ic & 31) * 8);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * NT + tid;
                if (i < N1) *(uint4*)(w1l + (i >> 5) * W1P + (i & 31) * 16) = v[u];
            }
        }
    }
    for (int i = tid; i < K; i += WAVES * 64) {
        sct[i] = p.scale3 ? p.scale3[i] : 1.f;
        sct[K + i] = p.shift3 ? p.shift3[i] : 0.f;
    }
    for (int tid2 = tid; tid2 < N2; tid2 += WAVES * 64) {
        const int tid = tid2;
        sct[2 * K + tid] = p.scale1 ? p.scale1[tid] : 1.f;
        sct[2 * K + N2 + tid] = p.shift1 ? p.shift1[tid] : 0.f;
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const int gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
    const char* w3f = w3l + fr * W3P + fh * 16;                 // + (a*32)*W3P + kk*32
    const char* w1f = w1l + fr * W1P + fh * 16;                 // + (a2*32)*W1P + (chunk*4 + kk2)*32

    auto load_x = [&](uint4* xf, int tile) {
        int m = tile * 32 + fr;
        m = m < p.M ? m : p.M - 1;                              // clamp: rows past the end are never stored
        const bf16_t* src = p.x + (long long)m * 64 + fh * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xf[kk] = *(const uint4*)(src + kk * 16);
        if constexpr (DUAL) {
            const bf16_t* src2 = p.x2 + (long long)m * 64 + fh * 8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xf[4 + kk] = *(const uint4*)(src2 + kk * 16);
        }
    };

    auto run_tile = [&](uint4* xf, int tile, int refill) {
        // stores through buffer descriptors at the tile's first row: a row past M gets an out-of-range offset instead of an `if` --
        // with the `if`s hipcc lost count of the in-order vmcnt queue and drained it (`vmcnt(0)`) in the middle of every tile, i.e.
        // waited for the NEXT tiles' x rows that had just been issued to hide their latency (round 5, tools/scan_store_waits.py)
        const int tile_u = __builtin_amdgcn_readfirstlane(tile);
        const brsrc_t ry = SUB ? make_brsrc(p.y) : make_brsrc(p.y + (long long)tile_u * 32 * K, p.y != nullptr);
        const brsrc_t rt = make_brsrc(p.t1 + (long long)tile_u * 32 * N2);
        const int rows_left = p.M - tile_u * 32;
        unsigned ysub[4] = {0, 0, 0, 0};                        // SUB: byte offset of the lane's row of pass i in the compact y, or OOB
        if constexpr (SUB) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int m = tile * 32 + pass * 8 + (lane >> 3);
                const int hw = p.subH * p.subW;
                const int b = m / hw, rem = m - b * hw;
                const int h = rem / p.subW, w = rem - h * p.subW;
                const bool keep = m < p.M && !((h | w) & 1);
                ysub[pass] = keep ? (unsigned)(((b * (p.subH >> 1) + (h >> 1)) * (p.subW >> 1) + (w >> 1)) * K) * 2u : BUF_OOB;
            }
        }
        f32x16 acc2[T2];
#pragma unroll
        for (int a = 0; a < T2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[a][e] = 0.f;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {                        // 128-channel slab of y (unrolled: every vmcnt wait is then an exact count)
            uint4 rr[DUAL ? 1 : 2][4];                          // residual rows of the slab's two 64-channel chunks
            if constexpr (!DUAL) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int pass = 0; pass < 4; ++pass) {
                        int m = tile * 32 + pass * 8 + (lane >> 3);
                        m = m < p.M ? m : p.M - 1;
                        rr[c][pass] = *(const uint4*)(p.residual + (long long)m * K + ps * 128 + c * 64 + (lane & 7) * 8);
                    }
            }
            f32x16 acc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
            const char* wf = w3f + (size_t)ps * 128 * W3P;
#pragma unroll
            for (int kk = 0; kk < KC; ++kk)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const uint4 av = *(const uint4*)(wf + a * 32 * W3P + kk * 32);
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                                     __builtin_bit_cast(bf16x8, xf[kk]), acc[a], 0, 0, 0);
                }
            if (ps == 1) load_x(xf, refill);                      // unconditional (rows past M are clamped inside): a branch here costs the count
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int chunk = ps * 2 + c;                   // 64-channel chunk of y
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int a = 2 * c + a2;
                        const int nl = a2 * 32 + 8 * g + 4 * fh;
                        *(float4*)(ep + fr * EPITCH + nl * 4) =
                            make_float4(acc[a][4 * g], acc[a][4 * g + 1], acc[a][4 * g + 2], acc[a][4 * g + 3]);
                    }
                const int c8 = lane & 7;
                const int nloc = chunk * 64 + c8 * 8;
                const float4 s0 = *(const float4*)(sct + nloc), s1 = *(const float4*)(sct + nloc + 4);
                const float4 h0 = *(const float4*)(sct + K + nloc), h1 = *(const float4*)(sct + K + nloc + 4);
                const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const float shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int row = pass * 8 + (lane >> 3);
                    const int m = tile * 32 + row;
                    const float4 lo = *(const float4*)(ep + row * EPITCH + c8 * 32);
                    const float4 hi = *(const float4*)(ep + row * EPITCH + c8 * 32 + 16);
                    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    if constexpr (DUAL) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], scv[e], shv[e]);
                    } else {
                        const uint32_t rw[4] = {rr[c][pass].x, rr[c][pass].y, rr[c][pass].z, rr[c][pass].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v[2 * e] = fmaf(v[2 * e], scv[2 * e], shv[2 * e]) + __uint_as_float(rw[e] << 16);
                            v[2 * e + 1] = fmaf(v[2 * e + 1], scv[2 * e + 1], shv[2 * e + 1]) + __uint_as_float(rw[e] & 0xffff0000u);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    uint4 u;
                    u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
                    if constexpr (SUB) buf_store_u4(ry, ysub[pass] == BUF_OOB ? BUF_OOB : ysub[pass] + (unsigned)nloc * 2u, u);
                    else buf_store_u4(ry, row < rows_left ? (unsigned)(row * K + nloc) * 2u : BUF_OOB, u);
                    // the same bf16 values, row-major in the patch: the B operand of the second GEMM.  (This pass's
                    // fp32 reads of these rows are older LDS operations of the same wave: in-order, no hazard.)
                    *(uint4*)(ep + row * EPITCH + c8 * 16) = u;
                }
#pragma unroll
                for (int kk2 = 0; kk2 < 4; ++kk2) {
                    const uint4 bv = *(const uint4*)(ep + fr * EPITCH + (2 * kk2 + fh) * 16);
#pragma unroll
                    for (int a2 = 0; a2 < T2; ++a2) {
                        const uint4 av = *(const uint4*)(w1f + a2 * 32 * W1P + (chunk * 4 + kk2) * 32);
                        acc2[a2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av),
                                                                           __builtin_bit_cast(bf16x8, bv), acc2[a2], 0, 0, 0);
                    }
                }
            }
        }
        // ---- second layer's epilogue: 32 pixels x N2 channels, 64 at a time
#pragma unroll
        for (int c2 = 0; c2 < N2 / 64; ++c2) {
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nl = a2 * 32 + 8 * g + 4 * fh;
                    *(float4*)(ep + fr * EPITCH + nl * 4) =
                        make_float4(acc2[2 * c2 + a2][4 * g], acc2[2 * c2 + a2][4 * g + 1], acc2[2 * c2 + a2][4 * g + 2],
                                    acc2[2 * c2 + a2][4 * g + 3]);
                }
            const int c8 = lane & 7;
            const float* s2 = sct + 2 * K + c2 * 64 + c8 * 8;
            const float4 s0 = *(const float4*)s2, s1 = *(const float4*)(s2 + 4);
            const float4 h0 = *(const float4*)(s2 + N2), h1 = *(const float4*)(s2 + N2 + 4);
            const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = pass * 8 + (lane >> 3);
                const int m = tile * 32 + row;
                const float4 lo = *(const float4*)(ep + row * EPITCH + c8 * 32);
                const float4 hi = *(const float4*)(ep + row * EPITCH + c8 * 32 + 16);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], scv[e], shv[e]), 0.f);
                uint4 u;
                u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
                buf_store_u4(rt, row < rows_left ? (unsigned)(row * N2 + c2 * 64 + c8 * 8) * 2u : BUF_OOB, u);
            }
        }
    };

    uint4 xa[KC], xb[KC];                                       // two pixel tiles in flight per wave
    int tile = gw;
    // (Round 6 tried running both tiles of an iteration unconditionally -- rows clamped, stores out of range -- so that hipcc's
    // waits at the loop head become exact counts instead of vmcnt(11) .. vmcnt(0): 7 -> 3-4 drains per kernel in the ISA, and the
    // kernels got 4-8 % SLOWER alone, because a wave with an odd number of tiles then computes a whole tile for nothing
    // (12 544 tiles over 2 048 waves = 6.1 each: 8 instead of 7); profiles/r06/chain_loop_unconditional_second_tile_ab.txt.)
    if (tile < p.tiles_m) load_x(xa, tile);
    if (tile + nw < p.tiles_m) load_x(xb, tile + nw);
    for (; tile < p.tiles_m; tile += 2 * nw) {
        run_tile(xa, tile, tile + 2 * nw);
        if (tile + nw < p.tiles_m) run_tile(xb, tile + nw, tile + 3 * nw);
    }
}

int chain1x1_supported(long long M, int C, int K, int N2, int dtype) {
    return dtype == MV_BF16 && C == 64 && K == 256 && (N2 == 64 || N2 == 128) && M >= 8192 && M < (1LL << 31) - (1 << 20) &&
           !get_flag("no_chain");
}
int chain1x1_dual_supported(long long M, int C1, int C2, int K, int N2, int dtype) {
    return dtype == MV_BF16 && C1 == 64 && C2 == 64 && K == 256 && N2 == 64 && M >= 8192 && M < (1LL << 31) - (1 << 20) &&
           !get_flag("no_chain") && !get_flag("no_dual_chain");
}

template <int N2, int WAVES, bool DUAL, bool SUB = false>
static int chain_go(ChainP& p, hipStream_t st) {
    constexpr int SMEM = 256 * ((DUAL ? 128 : 64) * 2 + 16) + N2 * 528 + (2 * 256 + 2 * N2) * 4 + WAVES * 32 * (64 * 4 + 16);
    static_assert(SMEM <= 160 * 1024, "LDS");
    int gx = 256;
    const int need = (p.tiles_m + WAVES - 1) / WAVES;
    if (gx > need) gx = need;
    auto kern = chain1x1_kernel<N2, WAVES, DUAL, SUB>;
    static LdsAttrSite attr;
    MV_HIP(attr.ensure((const void*)kern, SMEM));
    hipLaunchKernelGGL(kern, dim3(gx), dim3(WAVES * 64), SMEM, st, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int chain1x1_launch(const void* x, const void* w3, const float* scale3, const float* shift3, const void* residual, void* y,
                    const void* w1, const float* scale1, const float* shift1, void* t1, long long M, int N2, hipStream_t st) {
    ChainP p;
    p.x = (const bf16_t*)x; p.x2 = nullptr; p.w3 = (const bf16_t*)w3; p.scale3 = scale3; p.shift3 = shift3;
    p.residual = (const bf16_t*)residual; p.y = (bf16_t*)y;
    p.w1 = (const bf16_t*)w1; p.scale1 = scale1; p.shift1 = shift1; p.t1 = (bf16_t*)t1;
    p.M = (int)M;
    p.tiles_m = (int)((M + 31) / 32);
    p.subH = p.subW = 0;
    if (N2 == 64) {
        set_kernel_name("chain1x1_bf16_64_256_64");
        return chain_go<64, 8, false>(p, st);
    }
    set_kernel_name("chain1x1_bf16_64_256_128");
    return chain_go<128, 6, false>(p, st);
}

int chain1x1_sub_supported(long long N, int H, int W, int C, int K, int N2, int dtype) {
    return chain1x1_supported(N * H * W, C, K, N2, dtype) && N2 == 128 && H % 2 == 0 && W % 2 == 0 &&
           N * (H / 2) * (W / 2) * K * 2 < (1LL << 31) && !get_flag("no_chain_sub");
}

// as chain1x1_launch with N2 = 128, y = the stride-2 sub-sampling [N][H/2][W/2][256] of the block output
int chain1x1_sub_launch(const void* x, const void* w3, const float* scale3, const float* shift3, const void* residual, void* y_sub,
                        const void* w1, const float* scale1, const float* shift1, void* t1, int N, int H, int W, hipStream_t st) {
    ChainP p;
    p.x = (const bf16_t*)x; p.x2 = nullptr; p.w3 = (const bf16_t*)w3; p.scale3 = scale3; p.shift3 = shift3;
    p.residual = (const bf16_t*)residual; p.y = (bf16_t*)y_sub;
    p.w1 = (const bf16_t*)w1; p.scale1 = scale1; p.shift1 = shift1; p.t1 = (bf16_t*)t1;
    const long long M = (long long)N * H * W;
    p.M = (int)M;
    p.tiles_m = (int)((M + 31) / 32);
    p.subH = H; p.subW = W;
    set_kernel_name("chain1x1_bf16_64_256_128_ysub2");
    return chain_go<128, 6, false, true>(p, st);
}

// x [M][64] and x2 [M][64] are the two K-sources, wcat [256][128] = [rows of W3 scaled | rows of W_d scaled], shift3 = the
// sum of the two folded BatchNorm shifts (scale3 may be NULL = 1)
int chain1x1_dual_launch(const void* x, const void* x2, const void* wcat, const float* scale3, const float* shift3, void* y,
                         const void* w1, const float* scale1, const float* shift1, void* t1, long long M, hipStream_t st) {
    ChainP p;
    p.x = (const bf16_t*)x; p.x2 = (const bf16_t*)x2; p.w3 = (const bf16_t*)wcat; p.scale3 = scale3; p.shift3 = shift3;
    p.residual = nullptr; p.y = (bf16_t*)y;
    p.w1 = (const bf16_t*)w1; p.scale1 = scale1; p.shift1 = shift1; p.t1 = (bf16_t*)t1;
    p.M = (int)M;
    p.tiles_m = (int)((M + 31) / 32);
    p.subH = p.subW = 0;
    set_kernel_name(y ? "chain1x1_dual_bf16_64+64_256_64" : "chain1x1_dual_bf16_64+64_256_64_noy");
    return chain_go<64, 6, true>(p, st);
}

}  // namespace mv

// ResNet-50 layer 1, second bottleneck boundary, WITHOUT the first block's output in HBM (round 6).
//
// chain1x1.hip fuses the last convolution of bottleneck i with the first one of bottleneck i+1, so the 256-channel block output y_i
// is written once and read once (as block i+1's identity).  Those two passes over y are still 40 % of layer 1's bytes -- and y_0 is
// cheap to RECOMPUTE: it is a pointwise function of two 64-channel maps that block 1's kernel can read instead (reference
// resnet.py:144-162, 295-303):
//
//   y0[m] = relu( [s3 W3_0 | sd Wd] . [t2_0[m] | x0[m]] + shift0 )                  (block 0: conv3 + downsample conv, one GEMM)
//   y1[m] = relu( scale1 * (W3_1 . t2_1[m]) + shift1 + bf16(y0[m]) )                -> HBM (block 2's identity)
//   t1[m] = relu( scaleN * (W1_2 . bf16(y1[m])) + shiftN )                          -> HBM (block 2's conv2 input)
//
// Block 0's kernel (chain1x1_dual with y = NULL) then stores only its t1, and this kernel reads t2_1, t2_0, x0 (3 x 64 channels)
// instead of t2_1 + y0 (64 + 256): -205 MB written and -103 MB read per 128 images.  The price is matrix work the HBM-bound chain
// kernels have to spare (their matrix pipes are 11-15 % busy): 16 MFMAs per 32 x 32 output chunk instead of 8.
//
// Structure: chain1x1's free-running waves over 32-pixel tiles (x fragments straight from HBM, two tiles in flight per wave, no block
// barrier in steady state), with three differences that keep LDS for the 128 KB of weights:
//   * weights sit in LDS in MFMA FRAGMENT ORDER (host: ops.chain_rc_fragments): fragment f = 64 lanes x 16 bytes, so the copy in is
//     linear and every A operand is one conflict-free ds_read_b128 at a compile-time offset;
//   * the whole chain stays in the ACCUMULATOR layout: y0's chunk, the residual add and y1's chunk are lane-local (a lane holds 4
//     consecutive channels of one pixel per accumulator quad in all three), and y1's bf16 chunk IS the B operand of the next conv1
//     once the host has permuted that layer's reduction index to the accumulator's channel order
//     (k-slot 8 fh + i  <->  channel 8 (2 s + i / 4) + 4 fh + i % 4): no LDS transpose between the GEMMs;
//   * LDS patches are bf16 and only serve the row-major stores (32 rows x 128 bytes): 4.5 KB per wave instead of 8.5.
#include "mfma_common.h"

namespace mv {

struct ChainRcP {
    const bf16_t* t2;      // [M][64] conv2 output of THIS block
    const bf16_t* t2p;     // [M][64] conv2 output of the previous (first) block
    const bf16_t* x0;      // [M][64] the stage input (the first block's input)
    const bf16_t* wf;      // 128 fragments of 1 KB (ops.chain_rc_fragments)
    const float* tab;      // shift0[256] scale1[256] shift1[256] scaleN[64] shiftN[64]
    bf16_t* y;             // [M][256]
    bf16_t* t1;            // [M][64]
    int M, tiles_m;
};

// NPREV = 1: the kernel described above.  NPREV = 0 (round 6): the FIRST boundary of the stage when its output map is not wanted --
//   t1[m] = relu( scaleN * (W1_1 . bf16(relu([s3 W3_0 | sd Wd] . [t2_0[m] | x0[m]] + shift0))) + shiftN )
// i.e. mv_conv1x1_dual_chain_fwd without y, rebuilt in this file's style: 96 KB of fragments, nothing staged for a store, 138 VGPRs
// -- so TWELVE waves per CU (three per SIMD) instead of six hide the LDS / HBM round trips of each other's tiles: 50.6 us per 128
// images against 88.2 for chain1x1_dual with y = NULL (profiles/r06/resnet50_layer1_recompute_plan_ab.txt).
// (NPREV = 1 with eight waves -- 32-channel store patches, 64-byte row pieces -- was built and is slower: 126 vs 118 us.  That
//  kernel is not short of waves: per tile its 128 MFMAs, ~1 200 VALU instructions and ~250 KB of LDS reads -- 128 KB of fragments,
//  96 KB of broadcast table reads -- each cost 30-50 us per launch, and they add up.)
template <int NPREV, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void chain_rc_kernel(const ChainRcP p) {
    constexpr int K = 256, N2 = 64, FPC = 12 + 4 * NPREV, NFRAG = 8 * FPC, PITCH = 144, TABN = (1 + 2 * NPREV) * K + 2 * N2,
                  NT = WAVES * 64, NX = 8 + 4 * NPREV, TN = (1 + 2 * NPREV) * K;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;
    float* tab = (float*)(smem + NFRAG * 1024);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* ep = (char*)(tab + TABN) + wave * (32 * PITCH);

    {   // weights -> LDS: a straight copy, several loads in flight per thread
        constexpr int N16 = NFRAG * 64, U = 4;
        for (int base = 0; base < N16; base += U * NT) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * NT + tid;
                v[u] = ((const uint4*)p.wf)[i < N16 ? i : N16 - 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * NT + tid;
                if (i < N16) ((uint4*)wl)[i] = v[u];
            }
        }
        for (int i = tid; i < TABN; i += NT) tab[i] = p.tab[i];
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    typedef __attribute__((address_space(3))) const char* lds_cp;
    // LDS bases as OPAQUE 32-bit addresses: every read below is then base + compile-time immediate.  (Left to itself hipcc
    // materialised one address register per table read -- 96 of them -- and spilled them; scratch reloads count in vmcnt and put
    // `s_waitcnt vmcnt(0)` between the MFMAs.)
    unsigned wbase0 = (unsigned)(uintptr_t)(lds_cp)wl + lane * 16;             // fragments 0 .. 63
    unsigned wbase1 = wbase0 + 65536u;                                            // fragments 64 .. 127 (NPREV = 1)
    unsigned tbase = (unsigned)(uintptr_t)(lds_cp)(const char*)tab + fh * 16;   // lane's 4 channels of a quad: + (32 c + 8 g) * 4
    asm volatile("" : "+v"(wbase0), "+v"(wbase1), "+v"(tbase));
    auto afrag = [&](int f) -> bf16x8 {
        const lds_cp b = (lds_cp)(uintptr_t)(f < 64 ? wbase0 : wbase1);
        return __builtin_bit_cast(bf16x8, *(const __attribute__((address_space(3))) u32x4_t*)(b + (f & 63) * 1024));
    };
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    auto tabq = [&](int word) -> float4 {                        // 4 floats at table word `word` + 4 fh
        const f32x4v v = *(const __attribute__((address_space(3))) f32x4v*)((lds_cp)(uintptr_t)tbase + word * 4);
        return make_float4(v[0], v[1], v[2], v[3]);
    };

    // xf[0..3] = t2_0, xf[4..7] = x0 (the two K-sources of block 0's GEMM, in Wcat's column order), xf[8..11] = t2_1 (NPREV = 1)
    auto load_x = [&](uint4* xf, int tile) {
        int m = tile * 32 + fr;
        m = m < p.M ? m : p.M - 1;                               // clamp: rows past the end are never stored
        const long long off = (long long)m * 64 + fh * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xf[kk] = *(const uint4*)(p.t2p + off + kk * 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xf[4 + kk] = *(const uint4*)(p.x0 + off + kk * 16);
        if constexpr (NPREV == 1) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xf[8 + kk] = *(const uint4*)(p.t2 + off + kk * 16);
        }
    };

    auto run_tile = [&](uint4* xf, int tile, int refill) {
        const int tile_u = __builtin_amdgcn_readfirstlane(tile);
        const brsrc_t ry = make_brsrc(p.y + (long long)tile_u * 32 * K, NPREV == 1);
        const brsrc_t rt = make_brsrc(p.t1 + (long long)tile_u * 32 * N2);
        const int rows_left = p.M - tile_u * 32;
        f32x16 acc2[2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[a][e] = 0.f;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {                         // 64-channel slab of y1 = two 32-channel chunks
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int c = ps * 2 + cc;
                f32x16 a0, a1;
#pragma unroll
                for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {                 // the two independent accumulators interleaved
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag(c * FPC + kk), __builtin_bit_cast(bf16x8, xf[kk]), a0, 0, 0, 0);
                    if constexpr (NPREV == 1)
                        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag(c * FPC + 8 + kk), __builtin_bit_cast(bf16x8, xf[8 + kk]), a1, 0, 0, 0);
                }
#pragma unroll
                for (int kk = 4; kk < 8; ++kk)
                    a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag(c * FPC + kk), __builtin_bit_cast(bf16x8, xf[kk]), a0, 0, 0, 0);
                if (c == 7) load_x(xf, refill);                  // unconditional (rows clamped): a branch here costs the vmcnt count
                uint32_t pk[8];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 h0 = tabq(c * 32 + 8 * g);
                    const float h0v[4] = {h0.x, h0.y, h0.z, h0.w};
                    float r[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) r[j] = fmaxf(a0[4 * g + j] + h0v[j], 0.f);
                    const uint32_t w0 = pack_bf2(r[0], r[1]), w1 = pack_bf2(r[2], r[3]);      // y0 as block 0 would have stored it
                    if constexpr (NPREV == 0) {
                        pk[2 * g] = w0;
                        pk[2 * g + 1] = w1;
                    } else {
                        const float4 s1 = tabq(K + c * 32 + 8 * g), h1 = tabq(2 * K + c * 32 + 8 * g);
                        const float s1v[4] = {s1.x, s1.y, s1.z, s1.w}, h1v[4] = {h1.x, h1.y, h1.z, h1.w};
                        float v[4];
                        r[0] = __uint_as_float(w0 << 16); r[1] = __uint_as_float(w0 & 0xffff0000u);
                        r[2] = __uint_as_float(w1 << 16); r[3] = __uint_as_float(w1 & 0xffff0000u);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(a1[4 * g + j], s1v[j], h1v[j]) + r[j], 0.f);
                        pk[2 * g] = pack_bf2(v[0], v[1]);
                        pk[2 * g + 1] = pack_bf2(v[2], v[3]);
                        // row-major staging for the store: row = pixel fr, 4 consecutive channels = 8 bytes
                        *(uint2*)(ep + fr * PITCH + (cc * 32 + 8 * g + 4 * fh) * 2) = make_uint2(pk[2 * g], pk[2 * g + 1]);
                    }
                }
                // next block's conv1 on this chunk: the packed accumulator entries 8 s .. 8 s + 7 ARE the B operand of k-step s
                const uint4 b0 = make_uint4(pk[0], pk[1], pk[2], pk[3]), b1 = make_uint4(pk[4], pk[5], pk[6], pk[7]);
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2) {
                    acc2[a2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag(c * FPC + FPC - 4 + a2), __builtin_bit_cast(bf16x8, b0), acc2[a2], 0, 0, 0);
                    acc2[a2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afrag(c * FPC + FPC - 2 + a2), __builtin_bit_cast(bf16x8, b1), acc2[a2], 0, 0, 0);
                }
            }
            if constexpr (NPREV == 1) {
                wave_lds_fence();
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int row = pass * 8 + (lane >> 3);
                    const uint4 u = *(const uint4*)(ep + row * PITCH + (lane & 7) * 16);
                    buf_store_u4(ry, row < rows_left ? (unsigned)(row * K + ps * 64 + (lane & 7) * 8) * 2u : BUF_OOB, u);
                }
                wave_lds_fence();
            }
        }
        // ---- conv1 of the next block: 32 pixels x 64 channels
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = a2 * 32 + 8 * g + 4 * fh;
                const float4 sn = tabq(TN + a2 * 32 + 8 * g), hn = tabq(TN + N2 + a2 * 32 + 8 * g);
                const float snv[4] = {sn.x, sn.y, sn.z, sn.w}, hnv[4] = {hn.x, hn.y, hn.z, hn.w};
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = fmaxf(fmaf(acc2[a2][4 * g + j], snv[j], hnv[j]), 0.f);
                *(uint2*)(ep + fr * PITCH + ch * 2) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        wave_lds_fence();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3);
            const uint4 u = *(const uint4*)(ep + row * PITCH + (lane & 7) * 16);
            buf_store_u4(rt, row < rows_left ? (unsigned)(row * N2 + (lane & 7) * 8) * 2u : BUF_OOB, u);
        }
        wave_lds_fence();
    };

    uint4 xa[NX], xb[NX];                                       // two pixel tiles in flight per wave
    const int gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
    int tile = gw;
    if (tile < p.tiles_m) load_x(xa, tile);
    if (tile + nw < p.tiles_m) load_x(xb, tile + nw);
    for (; tile < p.tiles_m; tile += 2 * nw) {
        run_tile(xa, tile, tile + 2 * nw);
        if (tile + nw < p.tiles_m) run_tile(xb, tile + nw, tile + 3 * nw);
    }
}

int chain_rc_supported(long long M, int C, int K, int N2, int dtype) {
    return dtype == MV_BF16 && C == 64 && K == 256 && N2 == 64 && M >= 8192 && M < (1LL << 31) - (1 << 20) && !get_flag("no_chain") &&
           !get_flag("no_chain_rc");
}

template <int NPREV, int WAVES>
static int chain_rc_go(ChainRcP& p, hipStream_t st) {
    constexpr int SMEM = 8 * (12 + 4 * NPREV) * 1024 + ((1 + 2 * NPREV) * 256 + 2 * 64) * 4 + WAVES * 32 * 144;
    static_assert(SMEM <= 160 * 1024, "LDS");
    int gx = 256;
    const int need = (p.tiles_m + WAVES - 1) / WAVES;
    if (gx > need) gx = need;
    auto kern = chain_rc_kernel<NPREV, WAVES>;
    static LdsAttrSite attr;
    MV_HIP(attr.ensure((const void*)kern, SMEM));
    hipLaunchKernelGGL(kern, dim3(gx), dim3(WAVES * 64), SMEM, st, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int chain_rc_launch(const void* t2, const void* t2_prev, const void* x0, const void* wfrag, const float* tab, void* y, void* t1,
                    long long M, hipStream_t st) {
    ChainRcP p;
    p.t2 = (const bf16_t*)t2; p.t2p = (const bf16_t*)t2_prev; p.x0 = (const bf16_t*)x0; p.wf = (const bf16_t*)wfrag; p.tab = tab;
    p.y = (bf16_t*)y; p.t1 = (bf16_t*)t1;
    p.M = (int)M;
    p.tiles_m = (int)((M + 31) / 32);
    set_kernel_name("chain_rc1_bf16_64x3_256_64");
    return chain_rc_go<1, 6>(p, st);
}

// the first boundary without its output map: t2 = the first block's conv2 output, x0 = the stage input
int chain_rc0_launch(const void* t2, const void* x0, const void* wfrag, const float* tab, void* t1, long long M, hipStream_t st) {
    ChainRcP p;
    p.t2 = nullptr; p.t2p = (const bf16_t*)t2; p.x0 = (const bf16_t*)x0; p.wf = (const bf16_t*)wfrag; p.tab = tab;
    p.y = nullptr; p.t1 = (bf16_t*)t1;
    p.M = (int)M;
    p.tiles_m = (int)((M + 31) / 32);
    set_kernel_name("chain_rc0_bf16_64x2_256_64");
    return chain_rc_go<0, 12>(p, st);      // 8 waves: the same on the model (87.1 vs 87.0 k img/s, tools/ab_flag.py chain_rc0_waves, round 6)
}

}  // namespace mv

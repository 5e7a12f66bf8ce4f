// The chained pair of pointwise layers of chain1x1.hip for the stages whose weights do NOT fit in LDS (ResNet-50 layer2:
// conv3 128 -> 512 of one bottleneck + conv1 512 -> 128 of the next, reference resnet.py:144-162 / :330-333; 2 x 131 KB of
// bf16 weights):
//
//   y [m, 0:K]  = relu( scale3 * (W3  t2[m, 0:C]) + shift3 + res[m, 0:K] )        -> HBM (the next block's identity)
//   t1[m, 0:N2] = relu( scale1 * (W1n y [m, 0:K]) + shift1 )                      -> HBM (the next block's conv2 input)
//
// Un-fused, y is written once and read twice; the second read (103 MB per 128 images at 28 x 28 x 512) is what this kernel saves:
// 360 -> 257 MB per block boundary.
//
// The weights are STREAMED through LDS in CH = 32-channel chunks of y: chunk c needs rows CH c .. of W3 (8.5 KB) and the same
// columns of W1n (10 KB).  All waves of a block work on the same chunk at the same time, each on its own 32-pixel tile
// (x fragments in registers, as in stream1x1 / chain1x1), so a chunk is fetched from L2 once per block and 256 pixels:
// 1.0 KB of L2 traffic per pixel against 2.5 KB of HBM traffic.  Two chunk buffers in LDS and two register sets: every thread
// fetches its 2 x 16 bytes of chunk c+2 at the top of chunk c and writes chunk c+1 (fetched a chunk earlier) to the other
// buffer after its MFMAs; ONE workgroup barrier per chunk (raw s_barrier behind lgkmcnt(0): the global stores / prefetches of
// the wave stay in flight across it).  The residual of a chunk is fetched two chunks ahead as well, across tiles.
// Per chunk and wave: 8 MFMAs of the first GEMM, the fused epilogue of chain1x1 (fp32 scale / shift, residual, relu, bf16,
// store, the same bf16 values back into the wave's patch as the B operand), 8 MFMAs of the second GEMM.
// The second GEMM sees y exactly as the next layer would read it from HBM: bit-identical to the un-fused pair.
//
// Every s_waitcnt vmcnt in the chunk loop must be exact, or each chunk pays a full memory round trip (measured: 4.6 us per
// chunk instead of 1): the register rings are renamed statically (loop unrolled by 4, no register copies of prefetched
// values) and nothing between a load and its use is conditional -- stores of rows past the end go to row M-1 instead, with
// the values row M-1 gets anyway (its x and residual rows are what the clamped loads returned), so y must not alias residual.
#include <type_traits>
#include "mfma_common.h"

namespace mv {

typedef unsigned int wq_t __attribute__((ext_vector_type(4)));   // a native vector: the chunk pieces must stay in registers

struct ChainSP {
    const bf16_t* x;        // t2 [M][C]
    const bf16_t* w3;       // [K][C]
    const float* scale3;
    const float* shift3;
    const bf16_t* residual; // [M][K]
    bf16_t* y;              // [M][K]
    const bf16_t* w1;       // [N2][K]
    const float* scale1;
    const float* shift1;
    bf16_t* t1;             // [M][N2]
    int M, tiles_m, rounds, act;   // act: waves of a block that take a tile in a round (the tiles of a block split evenly over its rounds)
};

template <int C, int K, int N2, int CH, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void chain_stream_kernel(const ChainSP p) {
    constexpr int KC = C / 16, NCH = K / CH, T2 = N2 / 32, UNR = 4;
    constexpr int W3P = C * 2 + 16;                             // odd number of 16-byte slots
    constexpr int W1P = CH * 2 + 16;
    constexpr int WB = CH * W3P + N2 * W1P;                     // one chunk buffer
    constexpr int EPITCH = CH * 4 + 16;                         // fp32 patch [32 px][CH]; the bf16 copy reuses it at BPITCH
    constexpr int BPITCH = CH * 2 + 16;
    constexpr int NT = WAVES * 64;
    constexpr int P3 = CH * (C / 8), P1 = N2 * (CH / 8), PT = (P3 + P1) / NT;
    constexpr int LPR = CH / 8, RPP = 64 / LPR, NPASS = 32 / RPP;   // epilogue: lanes per row, rows per pass, passes
    static_assert(CH == 32 && (P3 + P1) % NT == 0 && P3 % NT == 0 && NCH % UNR == 0, "chunk pieces must split evenly over the threads");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wbuf = smem;                                          // [2][WB]
    float* sct = (float*)(smem + 2 * WB);                       // scale3[K], shift3[K], scale1[N2], shift1[N2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* ep = (char*)(sct + 2 * K + 2 * N2) + wave * (32 * EPITCH);

    // my pieces of a weight chunk: pieces 0 .. P3-1 are W3 rows (C/8 pieces each), the rest W1n rows (CH/8 pieces each)
    auto wload = [&](int c, wq_t* wr) {
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int i = u * NT + tid;
            if (u * NT < P3) {
                const int row = i / (C / 8), col = i - row * (C / 8);
                wr[u] = *(const wq_t*)(p.w3 + (long long)(CH * c + row) * C + col * 8);
            } else {
                const int j = i - P3, row = j / (CH / 8), col = j - row * (CH / 8);
                wr[u] = *(const wq_t*)(p.w1 + (long long)row * K + CH * c + col * 8);
            }
        }
    };
    auto wstore = [&](int buf, const wq_t* wr) {
        char* b = wbuf + buf * WB;
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            const int i = u * NT + tid;
            if (u * NT < P3) {
                const int row = i / (C / 8), col = i - row * (C / 8);
                *(wq_t*)(b + row * W3P + col * 16) = wr[u];
            } else {
                const int j = i - P3, row = j / (CH / 8), col = j - row * (CH / 8);
                *(wq_t*)(b + CH * W3P + row * W1P + col * 16) = wr[u];
            }
        }
    };
    auto block_sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);                      // nothing of the next chunk is hoisted above the barrier
    };

    wq_t wreg[2][PT];                                           // chunk c+1 in wreg[(c + 1) & 1], chunk c+2 in wreg[c & 1]
    wload(0, wreg[0]);
    wload(1, wreg[1]);
    for (int i = tid; i < K; i += NT) {
        sct[i] = p.scale3 ? p.scale3[i] : 1.f;
        sct[K + i] = p.shift3 ? p.shift3[i] : 0.f;
    }
    for (int i = tid; i < N2; i += NT) {
        sct[2 * K + i] = p.scale1 ? p.scale1[i] : 1.f;
        sct[2 * K + N2 + i] = p.shift1 ? p.shift1[i] : 0.f;
    }
    wstore(0, wreg[0]);
    block_sync();

    const int fr = lane & 31, fh = lane >> 5;
    const int erow = lane / LPR, ecol = lane % LPR;             // epilogue: row within a pass, 8-channel group
    auto load_x = [&](uint4* xf, int tile) {
        int m = tile * 32 + fr;
        m = m < p.M ? m : p.M - 1;                              // clamp: a tile past the end recomputes row M-1
        const bf16_t* src = p.x + (long long)m * C + fh * 8;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) xf[kk] = *(const uint4*)(src + kk * 16);
    };
    auto load_res = [&](uint4* rr, int tile, int chunk) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            int m = tile * 32 + pass * RPP + erow;
            m = m < p.M ? m : p.M - 1;
            rr[pass] = *(const uint4*)(p.residual + (long long)m * K + chunk * CH + ecol * 8);
        }
    };

    uint4 xf[KC];
    // Block b owns tiles [tile0, tile0 + nb): T / G of them, one more for the first T % G blocks, so that all CUs finish
    // together; they are taken `act` at a time (the same number in every round), wave w < act taking tile r * act + w.
    const int tbase = p.tiles_m / (int)gridDim.x, trem = p.tiles_m % (int)gridDim.x;
    const int nb = tbase + ((int)blockIdx.x < trem ? 1 : 0);
    const int tile0 = (int)blockIdx.x * tbase + ((int)blockIdx.x < trem ? (int)blockIdx.x : trem);
    const int stride = p.act;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    int tile = tile0 + wave;
    load_x(xf, tile);
    uint4 rr[4][NPASS];                                         // the residual of chunk c in rr[c & 3], fetched two chunks ahead
    load_res(rr[0], tile, 0);
    load_res(rr[1], tile, 1);
    for (int r = 0; r < p.rounds; ++r, tile += stride) {
        if (!(wv < p.act && r * p.act + wv < nb)) {             // no tile for this wave (nor in any later round): keep the weights moving
#pragma unroll 1
            for (int c0 = 0; c0 < NCH; c0 += 2)
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    wload((c0 + ci + 2) % NCH, wreg[ci & 1]);
                    wstore((ci + 1) & 1, wreg[(ci + 1) & 1]);
                    block_sync();
                }
            continue;
        }
        f32x16 acc2[T2];
#pragma unroll
        for (int a = 0; a < T2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[a][e] = 0.f;
        // UNR chunks; `last`: the round's last group, which also fetches the next tile's x (peeled, not a branch: see above)
        auto group = [&](const int c0, auto last) {
#pragma unroll
        for (int ci = 0; ci < UNR; ++ci) {
            const int c = c0 + ci;
            const char* w3c = wbuf + (ci & 1) * WB + fr * W3P + fh * 16;             // + kk*32
            const char* w1c = wbuf + (ci & 1) * WB + CH * W3P + fr * W1P + fh * 16;  // + a2*32*W1P + kk2*32
            {                                                   // chunks c+2 (weights, residual): the next tile's once c+2 >= NCH
                const int cn = c + 2 < NCH ? c + 2 : c + 2 - NCH;
                wload(cn, wreg[ci & 1]);
                load_res(rr[(ci + 2) & 3], c + 2 < NCH ? tile : tile + stride, cn);
            }
            // ---- first GEMM: CH channels of y for my 32 pixels, straight into the wave's patch
            {
                f32x16 acc;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < KC; ++kk) {
                    const uint4 av = *(const uint4*)(w3c + kk * 32);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, xf[kk]),
                                                                  acc, 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4*)(ep + fr * EPITCH + (8 * g + 4 * fh) * 4) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            }
            if constexpr (decltype(last)::value)
                if (ci == UNR - 1) load_x(xf, tile + stride);   // next round's pixels: the fragments are free now
            // ---- its epilogue: patch -> row-major (scale / shift, residual, relu, bf16, store, B operand of the second GEMM)
            wave_lds_fence();
            const int nloc = c * CH + ecol * 8;
            const float4 s0 = *(const float4*)(sct + nloc), s1 = *(const float4*)(sct + nloc + 4);
            const float4 h0 = *(const float4*)(sct + K + nloc), h1 = *(const float4*)(sct + K + nloc + 4);
            const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            uint4 ub[NPASS];
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int row = pass * RPP + erow;
                int m = tile * 32 + row;
                m = m < p.M ? m : p.M - 1;                      // rows past the end rewrite row M-1 with its own values
                const float4 lo = *(const float4*)(ep + row * EPITCH + ecol * 32);
                const float4 hi = *(const float4*)(ep + row * EPITCH + ecol * 32 + 16);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const uint4 rq = rr[ci][pass];
                const uint32_t rw[4] = {rq.x, rq.y, rq.z, rq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = fmaxf(fmaf(v[2 * e], scv[2 * e], shv[2 * e]) + __uint_as_float(rw[e] << 16), 0.f);
                    v[2 * e + 1] = fmaxf(fmaf(v[2 * e + 1], scv[2 * e + 1], shv[2 * e + 1]) + __uint_as_float(rw[e] & 0xffff0000u), 0.f);
                }
                uint4 u;
                u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
                *(uint4*)(p.y + (long long)m * K + nloc) = u;
                ub[pass] = u;
            }
            wave_lds_fence();                                   // every lane has read its fp32 rows: the patch becomes the bf16 copy
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass)            // the same bf16 values, row-major: the B operand of the second GEMM
                *(uint4*)(ep + (pass * RPP + erow) * BPITCH + ecol * 16) = ub[pass];
            wave_lds_fence();
            // ---- second GEMM: N2 outputs += W1n[:, chunk] . y_chunk
#pragma unroll
            for (int kk2 = 0; kk2 < CH / 16; ++kk2) {
                const uint4 bv = *(const uint4*)(ep + fr * BPITCH + (2 * kk2 + fh) * 16);
#pragma unroll
                for (int a2 = 0; a2 < T2; ++a2) {
                    const uint4 av = *(const uint4*)(w1c + a2 * 32 * W1P + kk2 * 32);
                    acc2[a2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                                       acc2[a2], 0, 0, 0);
                }
            }
            wave_lds_fence();                                   // the patch is rewritten by the next chunk
            wstore((ci + 1) & 1, wreg[(ci + 1) & 1]);           // chunk c+1; that buffer was last read before the previous barrier
            block_sync();
        }
        };
#pragma unroll 1
        for (int c0 = 0; c0 < NCH - UNR; c0 += UNR) group(c0, std::false_type());
        group(NCH - UNR, std::true_type());
        // ---- second layer's epilogue: 32 pixels x N2 channels, 32 at a time
#pragma unroll
        for (int a2 = 0; a2 < T2; ++a2) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(ep + fr * EPITCH + (8 * g + 4 * fh) * 4) =
                    make_float4(acc2[a2][4 * g], acc2[a2][4 * g + 1], acc2[a2][4 * g + 2], acc2[a2][4 * g + 3]);
            wave_lds_fence();
            const float* s2 = sct + 2 * K + a2 * 32 + ecol * 8;
            const float4 s0 = *(const float4*)s2, s1 = *(const float4*)(s2 + 4);
            const float4 h0 = *(const float4*)(s2 + N2), h1 = *(const float4*)(s2 + N2 + 4);
            const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
            const float shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                const int row = pass * RPP + erow;
                int m = tile * 32 + row;
                m = m < p.M ? m : p.M - 1;
                const float4 lo = *(const float4*)(ep + row * EPITCH + ecol * 32);
                const float4 hi = *(const float4*)(ep + row * EPITCH + ecol * 32 + 16);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(fmaf(v[e], scv[e], shv[e]), 0.f);
                Out8<bf16_t>::st(p.t1 + (long long)m * N2 + a2 * 32 + ecol * 8, v);
            }
            wave_lds_fence();
        }
    }
}

int chain_stream_supported(long long M, int C, int K, int N2, int dtype) {
    return dtype == MV_BF16 && C == 128 && K == 512 && N2 == 128 && M >= 16384 && M < (1LL << 31) - 64 && !get_flag("no_chain") &&
           !get_flag("no_chain_stream");
}

int chain_stream_launch(const void* x, const void* w3, const float* scale3, const float* shift3, const void* residual, void* y,
                        const void* w1, const float* scale1, const float* shift1, void* t1, long long M, hipStream_t st) {
    constexpr int C = 128, K = 512, N2 = 128, CH = 32, WAVES = 8;
    constexpr int SMEM = 2 * (CH * (C * 2 + 16) + N2 * (CH * 2 + 16)) + (2 * K + 2 * N2) * 4 + WAVES * 32 * (CH * 4 + 16);
    static_assert(SMEM <= 160 * 1024, "LDS");
    ChainSP p;
    p.x = (const bf16_t*)x; p.w3 = (const bf16_t*)w3; p.scale3 = scale3; p.shift3 = shift3;
    p.residual = (const bf16_t*)residual; p.y = (bf16_t*)y;
    p.w1 = (const bf16_t*)w1; p.scale1 = scale1; p.shift1 = shift1; p.t1 = (bf16_t*)t1;
    p.M = (int)M;
    p.tiles_m = (int)((M + 31) / 32);
    int gx = 256;
    const int need = (p.tiles_m + WAVES - 1) / WAVES;
    if (gx > need) gx = need;
    const int per_block = (p.tiles_m + gx - 1) / gx;            // the largest share of a block
    p.rounds = (per_block + WAVES - 1) / WAVES;
    p.act = (per_block + p.rounds - 1) / p.rounds;
    set_kernel_name("chain_stream_bf16_128_512_128");
    auto kern = chain_stream_kernel<C, K, N2, CH, WAVES>;
    MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    hipLaunchKernelGGL(kern, dim3(gx), dim3(WAVES * 64), SMEM, st, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // namespace mv

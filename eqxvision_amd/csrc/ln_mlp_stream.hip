// The MLP half of a pre-norm transformer block in ONE launch for rows too wide for the weights to live in LDS (gfx950):
//     y = x + fc2(gelu_tanh(fc1(LayerNorm(x))))            (swin.py:572-578 second line; mlps.py:54-66; extensions_2d.py:9-28)
// Swin stages 1-2 (C = 192 / 384, hidden = 4 C).  ln_mlp.hip keeps both weight matrices in LDS (C = 96); here they are STREAMED:
//
//   * a 512-thread block owns TM = 64 token rows: their LayerNorm (fp32 statistics) goes to LDS once as bf16 (rows padded by
//     16 bytes: conflict-free ds_read_b128 without a swizzle), the hidden activations go through LDS in chunks of 256 units
//     (two buffers, ONE barrier per chunk) and never reach HBM -- 77 MB per launch at stage 2 that the un-fused pair writes
//     and reads back;
//   * operands as in bneck_tail.hip: A = weights (rows = output units) come straight from L2 into registers, pre-arranged by
//     the host in fragment order (every load one coalesced 1 KB piece, rolling prefetch); B = token rows from LDS;
//   * fc1: a wave owns 32 hidden units of the chunk x all 64 rows; bias + GELU in registers, bf16 to the chunk buffer.
//     fc2: 12 output-channel tiles on 8 waves: wave w owns tile w over the whole reduction and ONE K-HALF of tile 8 + (w & 3)
//     (waves 0-3 the first 8 k-steps of a chunk, waves 4-7 the last 8) -- every wave runs the same 24 + 24 steps per chunk,
//     every weight fragment is fetched by exactly one wave (the per-CU L2 -> register path, ~64 bytes / clock, is what this
//     kernel leans on: 2.4 MB of weights per 64 rows at C = 384), and the two partial sums of an extra tile meet in the
//     epilogue's LDS tile;
//   * the first fragments of a phase are requested one phase ahead (their registers are free), so no phase starts on an L2
//     round trip; epilogue: the 64 x C fp32 result goes through LDS (free by then) so that the residual rows are re-read and
//     the output written as whole coalesced rows (from the accumulator layout a lane owns 4 channels of a row: 16-byte
//     pieces at a 1.5 KB stride took 13k cycles per block).
#include <type_traits>

#include "mfma_common.h"

namespace mv {

namespace {

struct LnMlpSP {
    const float* x;       // [M][C] fp32 residual stream
    const bf16_t* w1f;    // [HID/256][8 waves][C/16][64][8]: fc1 with the LayerNorm scale folded in
    const float* b1;      // [HID] (LayerNorm shift folded in)
    const bf16_t* w2f;    // [HID/256][C/32][16][64][8]
    const float* b2;      // [C]
    float* y;             // [M][C]
    long long M;
    float eps;
    long long* prof;      // experiments only (tools/time_ln_mlp_stream.py): per-wave wall-clock / shader-clock stamps
};

template <int C, int TM, int D1, int D2>
__global__ __launch_bounds__(512) void ln_mlp_stream_kernel(const LnMlpSP p) {
    constexpr int HID = 4 * C;
    constexpr int NCH = HID / 256;           // hidden chunks
    constexpr int KS1 = C / 16;              // k16-steps of fc1
    constexpr int KS2 = 16;                  // k16-steps of fc2 per chunk
    constexpr int NTB = TM / 32;             // token blocks
    constexpr int NCT = C / 32;              // output-channel tiles of fc2
    static_assert(NTB == 2 && NCT == 12 && C % 48 == 0, "C = 384 layout: 8 full tiles + 4 tiles in K-halves");
    static_assert(KS1 % D1 == 0 && KS2 % D2 == 0, "the rolling fragment buffers run through phase boundaries: depths must divide the step counts");
    constexpr int XROW = C * 2 + 16;
    constexpr int HROW = 256 * 2 + 16;
    constexpr int LDS_H = TM * XROW;
    // D1 / D2: fragments in flight per stream (an fc1 step is 2 MFMAs, an fc2 step 2-4: an L2 round trip is many steps)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const long long row0 = (long long)blockIdx.x * TM;
    long long st_w[8];
    unsigned st_c[8];
    int nst = 0;
#define MV_LMS_STAMP()                                                   \
    do {                                                                 \
        if (p.prof && nst < 8) {                                         \
            st_w[nst] = wall_clock64();                                  \
            st_c[nst] = (unsigned)__builtin_readcyclecounter();          \
            ++nst;                                                       \
        }                                                                \
    } while (0)
    MV_LMS_STAMP();

    // weight-fragment streams of this wave
    const uint4* ap1 = (const uint4*)p.w1f + (size_t)wave * KS1 * 64 + lane;                 // + chunk * 8 * KS1 * 64
    const int ct0 = wave, ct1 = 8 + (wave & 3);                                              // channel tiles of fc2: full, K-half
    const int h0 = (wave >> 2) * (KS2 / 2);                                                  // first k-step of this wave's half of ct1
    const uint4* ap2a = (const uint4*)p.w2f + (size_t)ct0 * KS2 * 64 + lane;                 // + chunk * NCT * KS2 * 64
    uint4 a1[D1], a2a[D2];
#pragma unroll
    for (int d = 0; d < D1; ++d) a1[d] = ap1[d * 64];

    // ---------------- LayerNorm of the tile -> LDS (bf16) -------------------------------------------------------------
    {
        constexpr int LPR = C / 12;                       // lanes per row: each holds 3 float4
        constexpr int RPP = 64 / LPR;                     // rows per wave pass
        constexpr int NP = 8 / RPP;                       // passes: a wave normalises 8 rows
        const int lr = lane / LPR, lq = lane % LPR;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int r = 8 * wave + ps * RPP + lr;
            long long gr = row0 + r;
            gr = gr < p.M ? gr : p.M - 1;
            const float4* src = (const float4*)(p.x + gr * C);
            float4 v[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) v[i] = src[lq + LPR * i];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float mean = s * (1.0f / C);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
            const float rstd = rsqrtf(q * (1.0f / C) + p.eps);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                uint2 u;
                u.x = pack_bf2(v[i].x * rstd, v[i].y * rstd);
                u.y = pack_bf2(v[i].z * rstd, v[i].w * rstd);
                *(uint2*)(smem + r * XROW + (lq + LPR * i) * 8) = u;
            }
        }
    }
#pragma unroll
    for (int d = 0; d < D2; ++d) a2a[d] = ap2a[d * 64];
    __syncthreads();

    f32x16 acc2[2][NTB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NTB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[a][b][e] = 0.f;

    const char* xb = smem + fr * XROW + fh * 16;                      // B fragments of fc1: + tb * 32 * XROW + j * 32
    const int hoff = fr * HROW + fh * 16;                             // B fragments of fc2: + tb * 32 * HROW + j * 32

    // Software pipeline over the hidden chunks (one barrier per chunk):
    //   P1(c): fc1 of chunk c -> 2 accumulator tiles                                        (matrix pipe)
    //   P2(c): bias + GELU of chunk c -> bf16 -> chunk buffer c & 1, issued PIECE BY PIECE between the MFMAs of fc2 of chunk
    //          c - 1 (which reads buffer (c - 1) & 1): the ~290 VALU / transcendental instructions of a chunk's GELU ride in the
    //          shadow of 32-64 MFMAs instead of running with the matrix pipe idle on both waves of the SIMD at once
    f32x16 acc1[NTB];
    float4 bia[4];
    bf16x8 bq[3][NTB];
    auto fc1 = [&](int c) {
#pragma unroll
        for (int b = 0; b < NTB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[b][e] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) bia[g] = *(const float4*)(p.b1 + c * 256 + 32 * wave + 8 * g + 4 * fh);
        const uint4* a1p = ap1 + (size_t)c * 8 * KS1 * 64;
        const uint4* a1n = ap1 + (size_t)(c + 1 < NCH ? c + 1 : c) * 8 * KS1 * 64;          // the next chunk's stream
        // position fragments run TWO steps ahead of their use (a step is only 2 MFMAs = 64 cycles of matrix work per wave:
        // one step ahead, the LDS round trip is exposed whenever the partner wave has nothing to issue)
#pragma unroll
        for (int b = 0; b < NTB; ++b) {
            bq[0][b] = *(const bf16x8*)(xb + b * 32 * XROW);
            bq[1][b] = *(const bf16x8*)(xb + b * 32 * XROW + 32);
        }
#pragma unroll
        for (int j = 0; j < KS1; ++j) {
            const int jn = j + 2 < KS1 ? j + 2 : KS1 - 1;
            // fragments D1 steps ahead; past the end of this chunk: the first fragments of the next chunk's fc1
            const uint4* an = (j + D1 < KS1) ? a1p + (size_t)(j + D1) * 64 : a1n + (size_t)(j + D1 - KS1) * 64;
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 af = __builtin_bit_cast(bf16x8, a1[j % D1]);
            a1[j % D1] = *an;
#pragma unroll
            for (int b = 0; b < NTB; ++b) {
                acc1[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[j % 3][b], acc1[b], 0, 0, 0);
                bq[(j + 2) % 3][b] = *(const bf16x8*)(xb + b * 32 * XROW + jn * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // bias + GELU -> bf16 -> the chunk buffer: row = token, 4 consecutive hidden units per store
    auto gelu_store = [&](int c) {
        char* hw = smem + LDS_H + (c & 1) * (TM * HROW);
#pragma unroll
        for (int b = 0; b < NTB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u;
                u.x = gelu_tanh_pack2(acc1[b][4 * g + 0] + bia[g].x, acc1[b][4 * g + 1] + bia[g].y);
                u.y = gelu_tanh_pack2(acc1[b][4 * g + 2] + bia[g].z, acc1[b][4 * g + 3] + bia[g].w);
                *(uint2*)(hw + (32 * b + fr) * HROW + (32 * wave + 8 * g + 4 * fh) * 2) = u;
            }
    };
    // fc2: a run of NS k-steps of ONE channel tile x 64 rows (hidden units 16 * k0 ... of chunk c).  A wave's runs form one
    // stream (chunk 0: tile w, 16 steps; tile 8 + (w & 3), its 8 steps; chunk 1: ...): the D2 fragments in flight run straight
    // through the run boundaries.  Every step is 1 fragment + 2 MFMAs, no conditional load inside a step (with one, the
    // compiler's vmcnt bookkeeping assumes the shorter queue and the wave waits on loads issued 1-2 steps earlier: 4x slower).
    // `with_gelu`: the bias + GELU + store of chunk `cg` (accumulators of the fc1 that has just run) is issued in 16 pieces
    // between the MFMAs of a 16-step run -- step j carries the pair (token block j / 8, channel group (j % 8) / 2, half j % 2).
    auto run_base = [&](int c, int second) -> const uint4* {           // fragment of the run's FIRST step
        c = c < NCH ? c : NCH - 1;
        return (const uint4*)p.w2f + ((size_t)(c * NCT + (second ? ct1 : ct0)) * KS2 + (second ? h0 : 0)) * 64 + lane;
    };
    auto fc2_run = [&](auto with_gelu, auto nsc, const uint4* cur, const uint4* nxt, int k0, f32x16 (&acc)[NTB], int c, int cg) {
        constexpr bool GELU = decltype(with_gelu)::value;
        constexpr int NS = decltype(nsc)::value;
        static_assert(NS % D2 == 0 && (!GELU || NS == 16), "run length");
        const char* hb = smem + LDS_H + (c & 1) * (TM * HROW) + hoff + k0 * 32;
        char* hw = smem + LDS_H + (cg & 1) * (TM * HROW);
#pragma unroll
        for (int b = 0; b < NTB; ++b) {
            bq[0][b] = *(const bf16x8*)(hb + b * 32 * HROW);
            bq[1][b] = *(const bf16x8*)(hb + b * 32 * HROW + 32);
        }
        uint32_t ulo = 0;
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int jn = j + 2 < NS ? j + 2 : NS - 1;
            const uint4* an = (j + D2 < NS) ? cur + (size_t)(j + D2) * 64 : nxt + (size_t)(j + D2 - NS) * 64;
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 af = __builtin_bit_cast(bf16x8, a2a[j % D2]);
            a2a[j % D2] = *an;
#pragma unroll
            for (int b = 0; b < NTB; ++b) {
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[j % 3][b], acc[b], 0, 0, 0);
                bq[(j + 2) % 3][b] = *(const bf16x8*)(hb + b * 32 * HROW + jn * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (GELU) {
                const int b = j >> 3, g = (j & 7) >> 1, hf = j & 1;
                const float bx = hf ? bia[g].z : bia[g].x, by = hf ? bia[g].w : bia[g].y;
                const uint32_t uu = gelu_tanh_pack2(acc1[b][4 * g + 2 * hf] + bx, acc1[b][4 * g + 2 * hf + 1] + by);
                if (hf) {
                    uint2 st;
                    st.x = ulo; st.y = uu;
                    *(uint2*)(hw + (32 * b + fr) * HROW + (32 * wave + 8 * g + 4 * fh) * 2) = st;
                } else {
                    ulo = uu;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    using N16 = std::integral_constant<int, KS2>;
    using N8 = std::integral_constant<int, KS2 / 2>;
    auto fc2_chunk = [&](auto with_gelu, int c, int cg) {              // both runs of chunk c
        fc2_run(with_gelu, N16{}, run_base(c, 0), run_base(c, 1), 0, acc2[0], c, cg);
        fc2_run(F_{}, N8{}, run_base(c, 1), run_base(c + 1, 0), h0, acc2[1], c, 0);
    };
    MV_LMS_STAMP();          // 1: LayerNorm done
    fc1(0);
    MV_LMS_STAMP();          // 2: fc1(0)
    gelu_store(0);
    __syncthreads();
    MV_LMS_STAMP();          // 3: GELU(0) + barrier
    for (int c = 1; c < NCH; ++c) {
        fc1(c);
        if (c == 1) MV_LMS_STAMP();      // 4: fc1(1)
        fc2_chunk(T_{}, c - 1, c);
        if (c == 1) MV_LMS_STAMP();      // 5: fc2(0) + GELU(1)
        __syncthreads();
        if (c == 1) MV_LMS_STAMP();      // 6: barrier
    }
    fc2_chunk(F_{}, NCH - 1, 0);
    MV_LMS_STAMP();          // 7: main loop done

    // ---------------- epilogue: accumulators -> LDS tile [64][C] fp32 (the two K-halves of an extra tile meet there), then
    // whole rows: + bias + residual row (coalesced fp32 re-read) -> coalesced stores ----------------------------------------
    constexpr int YROW = C * 4 + 16;
    static_assert(TM * YROW <= TM * XROW + 2 * TM * HROW, "result tile must fit in the (dead) operand buffers");
    __syncthreads();                                                   // nobody reads the operand buffers any more
    auto put = [&](const f32x16 (&acc)[NTB], int ct, bool add) {
#pragma unroll
        for (int b = 0; b < NTB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4* dst = (float4*)(smem + (32 * b + fr) * YROW + (32 * ct + 8 * g + 4 * fh) * 4);
                float4 v = make_float4(acc[b][4 * g + 0], acc[b][4 * g + 1], acc[b][4 * g + 2], acc[b][4 * g + 3]);
                if (add) {
                    const float4 o = *dst;
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                *dst = v;
            }
    };
    put(acc2[0], ct0, false);
    if (wave < 4) put(acc2[1], ct1, false);
    __syncthreads();
    if (wave >= 4) put(acc2[1], ct1, true);
    __syncthreads();
    {
        constexpr int QPR = C / 4;                                     // float4 per row
        constexpr int NIT = TM * QPR / 512;
        static_assert(TM * QPR % 512 == 0, "whole passes");
        // rows past M: out-of-range buffer offsets (loads return zeros, stores are dropped) -- no branch, no load inside the store
        // loop, so the NIT stores stream behind each other instead of one L2 round trip each (tools/scan_store_waits.py)
        const brsrc_t rx = make_brsrc(p.x + (long long)row0 * C), ry = make_brsrc(p.y + (long long)row0 * C), rb = make_brsrc(p.b2);
        const int rows_left = p.M - (int)row0;
        float4 xr[NIT], b2[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + 512 * i;
            const int r = idx / QPR, q = idx - r * QPR;
            xr[i] = buf_load_f4(rx, r < rows_left ? (unsigned)idx * 16u : BUF_OOB);
            b2[i] = buf_load_f4(rb, (unsigned)q * 16u);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + 512 * i;
            const int r = idx / QPR, q = idx - r * QPR;
            const float4 a = *(const float4*)(smem + r * YROW + q * 16);
            float4 o;
            o.x = a.x + b2[i].x + xr[i].x; o.y = a.y + b2[i].y + xr[i].y; o.z = a.z + b2[i].z + xr[i].z; o.w = a.w + b2[i].w + xr[i].w;
            buf_store_f4(ry, r < rows_left ? (unsigned)idx * 16u : BUF_OOB, o);
        }
    }
    if (p.prof) {
        const long long tend = wall_clock64();
        const unsigned cend = (unsigned)__builtin_readcyclecounter();
        if (lane == 0) {
            long long* o = p.prof + ((size_t)blockIdx.x * 8 + wave) * 18;
            for (int i = 0; i < 8; ++i) { o[i] = st_w[i]; o[9 + i] = (long long)st_c[i]; }
            o[8] = tend; o[17] = (long long)cend;
        }
    }
#undef MV_LMS_STAMP
}

// ---------------------------------------------------------------------------------------------------------------------------
// C = 192 (Swin stage 1): 128 rows per block, so that the 6 output-channel tiles x 4 token blocks of fc2 are 24 MFMA tiles, three
// per wave (the unit layout of swin_block_attn.hip), and every fc1 weight fragment feeds 4 MFMAs.  One chunk buffer (a second one
// does not fit next to 128 normalised rows): two barriers per chunk, fc1 of the next chunk runs ahead of the first.
template <int TM>
__global__ __launch_bounds__(512) void ln_mlp_stream192_kernel(const LnMlpSP p) {
    constexpr int C = 192, HID = 768, NCH = HID / 256, KS1 = C / 16, KS2 = 16, NTB = TM / 32, NCT = C / 32, D = 4, PB = 4;
    static_assert(NTB == 4 && NCT * NTB == 24 && KS1 % D == 0 && KS2 % D == 0 && KS1 % PB == 0 && KS2 % PB == 0, "layout");
    constexpr int XROW = C * 2 + 16, HROW = 256 * 2 + 16, LDS_H = TM * XROW, YROW = C * 4 + 16;
    static_assert(TM * YROW <= TM * XROW + TM * HROW, "result tile must fit in the (dead) operand buffers");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const long long row0 = (long long)blockIdx.x * TM;

    // fc2 units of this wave: 3w .. 3w+2 -> (tile = u / 4, token block = u % 4): at most two tiles
    const int ta = (3 * wave) >> 2;
    const int u_tile[3] = {(3 * wave) >> 2, (3 * wave + 1) >> 2, (3 * wave + 2) >> 2};
    const int u_tb[3] = {(3 * wave) & 3, (3 * wave + 1) & 3, (3 * wave + 2) & 3};
    const int pat = (u_tile[1] - ta) + 2 * (u_tile[2] - ta);
    auto t2_base = [&](int c, int tile) -> const uint4* {
        c = c < NCH ? c : NCH - 1;
        tile = tile < NCT ? tile : NCT - 1;
        return (const uint4*)p.w2f + ((size_t)(c * NCT + tile) * KS2) * 64 + lane;
    };
    const uint4* ap1 = (const uint4*)p.w1f + (size_t)wave * KS1 * 64 + lane;                 // + chunk * 8 * KS1 * 64
    uint4 a1[D], a20[D], a21[D];
#pragma unroll
    for (int d = 0; d < D; ++d) a1[d] = ap1[d * 64];

    // ---------------- LayerNorm of the tile -> LDS (bf16) ---------------------------------------------------------------------
    {
        constexpr int LPR = C / 12, RPP = 64 / LPR, RPW = TM / 8, NP = RPW / RPP;
        const int lr = lane / LPR, lq = lane % LPR;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int r = RPW * wave + ps * RPP + lr;
            long long gr = row0 + r;
            gr = gr < p.M ? gr : p.M - 1;
            const float4* src = (const float4*)(p.x + gr * C);
            float4 v[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) v[i] = src[lq + LPR * i];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
            const float mean = s * (1.0f / C);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
                q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
            }
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
            const float rstd = rsqrtf(q * (1.0f / C) + p.eps);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                uint2 u;
                u.x = pack_bf2(v[i].x * rstd, v[i].y * rstd);
                u.y = pack_bf2(v[i].z * rstd, v[i].w * rstd);
                *(uint2*)(smem + r * XROW + (lq + LPR * i) * 8) = u;
            }
        }
    }
    {
        const uint4* s0 = t2_base(0, ta);
        const uint4* s1 = t2_base(0, ta + 1);
#pragma unroll
        for (int d = 0; d < D; ++d) { a20[d] = s0[d * 64]; a21[d] = s1[d * 64]; }
    }
    __syncthreads();

    f32x16 acc2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[i][e] = 0.f;
    f32x16 acc1[NTB];
    float4 bia[4];
    const char* xb = smem + fr * XROW + fh * 16;
    char* hbuf = smem + LDS_H;

    auto fc1 = [&](int c) __attribute__((always_inline)) {
#pragma unroll
        for (int b = 0; b < NTB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[b][e] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) bia[g] = *(const float4*)(p.b1 + c * 256 + 32 * wave + 8 * g + 4 * fh);
        const uint4* a1p = ap1 + (size_t)c * 8 * KS1 * 64;
        const uint4* a1n = ap1 + (size_t)(c + 1 < NCH ? c + 1 : c) * 8 * KS1 * 64;
        bf16x8 bq[PB][NTB];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int b = 0; b < NTB; ++b) bq[t][b] = *(const bf16x8*)(xb + b * 32 * XROW + t * 32);
#pragma unroll
        for (int j = 0; j < KS1; ++j) {
            const int jn = j + 2 < KS1 ? j + 2 : KS1 - 1;
            const uint4* an = (j + D < KS1) ? a1p + (size_t)(j + D) * 64 : a1n + (size_t)(j + D - KS1) * 64;
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 af = __builtin_bit_cast(bf16x8, a1[j % D]);
            a1[j % D] = *an;
#pragma unroll
            for (int b = 0; b < NTB; ++b) {
                acc1[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bq[j % PB][b], acc1[b], 0, 0, 0);
                bq[(j + 2) % PB][b] = *(const bf16x8*)(xb + b * 32 * XROW + jn * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto fc2 = [&](auto patc, int c) __attribute__((always_inline)) {
        constexpr int PAT = decltype(patc)::value;
        const uint4* cur0 = t2_base(c, ta);
        const uint4* cur1 = t2_base(c, ta + 1);
        const uint4* nx0 = t2_base(c + 1, ta);
        const uint4* nx1 = t2_base(c + 1, ta + 1);
        const char* hb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) hb[i] = hbuf + (32 * u_tb[i] + fr) * HROW + fh * 16;
        bf16x8 bq[PB][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 3; ++i) bq[t][i] = *(const bf16x8*)(hb[i] + t * 32);
#pragma unroll
        for (int j = 0; j < KS2; ++j) {
            const int jn = j + 2 < KS2 ? j + 2 : KS2 - 1;
            const bool in = j + D < KS2;
            const uint4* n0 = in ? cur0 + (size_t)(j + D) * 64 : nx0 + (size_t)(j + D - KS2) * 64;
            const uint4* n1 = in ? cur1 + (size_t)(j + D) * 64 : nx1 + (size_t)(j + D - KS2) * 64;
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 f0 = __builtin_bit_cast(bf16x8, a20[j % D]);
            const bf16x8 f1 = __builtin_bit_cast(bf16x8, a21[j % D]);
            a20[j % D] = *n0;
            a21[j % D] = *n1;
            acc2[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, bq[j % PB][0], acc2[0], 0, 0, 0);
            acc2[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((PAT & 1) ? f1 : f0, bq[j % PB][1], acc2[1], 0, 0, 0);
            acc2[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((PAT & 2) ? f1 : f0, bq[j % PB][2], acc2[2], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i) bq[(j + 2) % PB][i] = *(const bf16x8*)(hb[i] + jn * 32);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    for (int c = 0; c < NCH; ++c) {
        fc1(c);
        __syncthreads();                     // every wave has finished reading the chunk buffer (fc2 of chunk c - 1)
#pragma unroll
        for (int b = 0; b < NTB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint2 u;
                u.x = gelu_tanh_pack2(acc1[b][4 * g + 0] + bia[g].x, acc1[b][4 * g + 1] + bia[g].y);
                u.y = gelu_tanh_pack2(acc1[b][4 * g + 2] + bia[g].z, acc1[b][4 * g + 3] + bia[g].w);
                *(uint2*)(hbuf + (32 * b + fr) * HROW + (32 * wave + 8 * g + 4 * fh) * 2) = u;
            }
        __syncthreads();
        if (pat == 0) fc2(std::integral_constant<int, 0>{}, c);
        else if (pat == 2) fc2(std::integral_constant<int, 2>{}, c);
        else fc2(std::integral_constant<int, 3>{}, c);
    }

    // ---------------- epilogue: accumulators -> LDS tile, then whole rows + bias + residual -----------------------------------------
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4*)(smem + (32 * u_tb[i] + fr) * YROW + (32 * u_tile[i] + 8 * g + 4 * fh) * 4) =
                make_float4(acc2[i][4 * g + 0], acc2[i][4 * g + 1], acc2[i][4 * g + 2], acc2[i][4 * g + 3]);
    __syncthreads();
    {
        constexpr int QPR = C / 4, NIT = TM * QPR / 512;
        static_assert(TM * QPR % 512 == 0, "whole passes");
        const brsrc_t rx = make_brsrc(p.x + (long long)row0 * C), ry = make_brsrc(p.y + (long long)row0 * C), rb = make_brsrc(p.b2);
        const int rows_left = p.M - (int)row0;
        float4 xr[NIT], b2[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + 512 * i;
            const int r = idx / QPR, q = idx - r * QPR;
            xr[i] = buf_load_f4(rx, r < rows_left ? (unsigned)idx * 16u : BUF_OOB);
            b2[i] = buf_load_f4(rb, (unsigned)q * 16u);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + 512 * i;
            const int r = idx / QPR, q = idx - r * QPR;
            const float4 a = *(const float4*)(smem + r * YROW + q * 16);
            buf_store_f4(ry, r < rows_left ? (unsigned)idx * 16u : BUF_OOB,
                         make_float4(a.x + b2[i].x + xr[i].x, a.y + b2[i].y + xr[i].y, a.z + b2[i].z + xr[i].z, a.w + b2[i].w + xr[i].w));
        }
    }
}


}  // namespace

}  // namespace mv

extern "C" {

int mv_ln_mlp_stream_supported(int64_t M, int C, int hidden, int x_dtype) {
    if (mv::get_flag("no_ln_mlp_stream")) return 0;
    return x_dtype == MV_F32 && (C == 384 || C == 192) && hidden == 4 * C && M >= 128;
}

int mv_ln_mlp_stream_fwd(const void* x, const void* w1f, const float* b1, const void* w2f, const float* b2, void* y, int64_t M,
                         int C, int hidden, float eps, int x_dtype, mv_stream_t stream_) {
    using namespace mv;
    hipStream_t stream = (hipStream_t)stream_;
    MV_CHECK_ARG(x && w1f && b1 && w2f && b2 && y, "mv_ln_mlp_stream_fwd: null argument");
    MV_CHECK_ARG(x != y, "mv_ln_mlp_stream_fwd: not in place");
    if (!mv_ln_mlp_stream_supported(M, C, hidden, x_dtype)) {
        set_error("mv_ln_mlp_stream_fwd: unsupported M=%lld C=%d hidden=%d (ask mv_ln_mlp_stream_supported first)", (long long)M, C, hidden);
        return MV_E_UNSUPPORTED;
    }
    LnMlpSP p;
    p.x = (const float*)x; p.w1f = (const bf16_t*)w1f; p.b1 = b1; p.w2f = (const bf16_t*)w2f; p.b2 = b2; p.y = (float*)y;
    p.M = M; p.eps = eps;
    p.prof = nullptr;
#ifdef MV_I8_PROF              // debug build only
    if (get_flag("lms_prof"))
        p.prof = (long long*)(((unsigned long long)(unsigned)get_flag("prof_hi") << 32) | (unsigned)get_flag("prof_lo"));
#endif
    if (C == 192) {
        constexpr int TM = 128;
        constexpr int SMEM = TM * (192 * 2 + 16) + TM * (256 * 2 + 16);
        auto kern = ln_mlp_stream192_kernel<TM>;
        MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
        set_kernel_name("ln_mlp_stream_c192_f32stream");
        hipLaunchKernelGGL(kern, dim3((unsigned)((M + TM - 1) / TM)), dim3(512), SMEM, stream, p);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    constexpr int TM = 64;
    constexpr int SMEM = TM * (384 * 2 + 16) + 2 * TM * (256 * 2 + 16);
    auto kern = ln_mlp_stream_kernel<384, TM, 8, 8>;         // prefetch depths: (4, 4), (6, 6), (8, 4) measured equal or slower
    MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    set_kernel_name("ln_mlp_stream_c384_f32stream");
    hipLaunchKernelGGL(kern, dim3((unsigned)((M + TM - 1) / TM)), dim3(512), SMEM, stream, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

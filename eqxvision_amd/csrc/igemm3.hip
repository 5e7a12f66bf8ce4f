// Phase-alternating ("ping-pong") implicit GEMM, 256 pixels x 256 channels per block, gfx950.
//
// igemm2.hip keeps all 8 waves in one schedule: every wave interleaves its own fragment reads and LDS-DMA
// issue with its own MFMAs.  PMC on the ViT Linears (profiles/r01): the matrix pipe is busy a third of the
// cycles and the waves sit in issue stalls 44% of the time -- a wave that is issuing a DMA piece
// (~100 cycles each, MI355X_MICROARCH.md) is not issuing MFMAs, and its SIMD partner is doing the same thing
// at the same time.
//
// Here the two waves of a SIMD never do the same thing:
//   * the 8 waves form two groups of four (group g = wave / 4 owns pixel rows 128 g .. 128 g + 127 of the tile;
//     its four waves own 64 channels each), one wave of each group per SIMD;
//   * a 32-wide k-tile is a PHASE with two halves separated by raw s_barriers:
//         LOAD  half: 12 ds_read_b128 (the tile's fragments), the wave's 4 LDS-DMA pieces of the k-tile three
//                     ahead, the counted vmcnt wait, lgkmcnt(0)
//         MFMA  half: 16 v_mfma_f32_32x32x16_bf16 (the LOAD half runs at s_setprio(2): measured better than
//                     boosting the MFMA half)
//     group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA half while the
//     other is in its LOAD half: the matrix pipe sees back-to-back bursts and nobody's loads sit in front
//     of anybody's MFMAs;
//   * k-tiles are 32 wide (64-byte rows), 4 LDS stages of 32 KB.  Tile u+3 is staged while tile u is
//     multiplied; its stage was last read by group 1 one barrier before the first piece is issued (WAR), and
//     `s_waitcnt vmcnt(8)` in the LOAD half of tile u plus the two barriers behind it order tile u+1's DMA
//     data before its first fragment read (RAW) -- never vmcnt(0) in the steady state.
//   * LDS rows are linear for the DMA (lane-linear destination); the 16-byte chunk a lane fetches is
//     XOR-swizzled on the SOURCE address (chunk ^ ((row >> 2) & 3)) and the fragment reads apply the same
//     XOR: ds_read_b128 is conflict-free for the hardware's 16-lane groups.
// Operands, accumulator layout and epilogue are igemm2's (A = weights, B = pixels: a lane ends up with 4
// consecutive channels of one pixel; LDS transpose, full 128-byte line stores).
#include "igemm_pipe.h"

namespace mv {

template <bool RESPF, typename OutT>
__global__ __launch_bounds__(512) void igemm3_kernel(const Igemm2P p) {
    constexpr int BM = 256, BN = 256, TM = 4, TN = 2;
    constexpr int ROWB = 64;                              // bytes per staged row: 32 bf16
    constexpr int NST = 4;
    constexpr int STAGE = (BM + BN) * ROWB;               // 32 KB
    constexpr int EPITCH = 64 * 4 + 16;
    static_assert(8 * 32 * EPITCH <= NST * STAGE, "epilogue patches must fit");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;                            // 0: leads, 1: one barrier behind
    const int t = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    int tile_m, tile_n;
    tile_coords(t, p.tiles_m, p.tiles_n, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---------------- DMA addressing: a piece = 16 rows x 64 bytes; wave w stages pieces w and w+8 of x and of w
    const int srow = lane >> 2;
    const int chunk = (lane & 3) ^ ((lane >> 4) & 3);     // source chunk for LDS slot lane&3 of row srow
    const int cpt = p.C >> 5;
    const int nk = p.R * p.S * cpt;
    const long long wrow_stride = (long long)p.R * p.S * p.C;
    const bool dense1x1 = p.R == 1 && p.S == 1 && p.sh == 1 && p.sw == 1 && p.ph == 0 && p.pw == 0;   // block-uniform
    long long xbase[2];
    unsigned vlo[2], vhi[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int m = m0 + 16 * (wave + 8 * j) + srow;
        const bool valid = m < p.M;
        if (dense1x1) {                                   // a Linear: row m is pixel m, one tap, no division chain
            xbase[j] = (long long)m * p.C + chunk * 8;
            vlo[j] = valid ? 1u : 0u;
            vhi[j] = 0u;
            continue;
        }
        const int wo = m % p.Wo;
        const int tt = m / p.Wo;
        const int ho = tt % p.Ho;
        const int b = tt / p.Ho;
        const int hi0 = ho * p.sh - p.ph, wi0 = wo * p.sw - p.pw;
        xbase[j] = (((long long)b * p.H + hi0) * p.W + wi0) * p.C + chunk * 8;
        unsigned long long mask = 0;
        if (valid) {
            for (int r = 0; r < p.R; ++r) {
                const int hi = hi0 + r * p.dh;
                if ((unsigned)hi >= (unsigned)p.H) continue;
                for (int s = 0; s < p.S; ++s) {
                    const int wi = wi0 + s * p.dw;
                    if ((unsigned)wi < (unsigned)p.W) mask |= 1ull << (r * p.S + s);
                }
            }
        }
        vlo[j] = (unsigned)mask;
        vhi[j] = (unsigned)(mask >> 32);
    }
    long long woff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + 16 * (wave + 8 * j) + srow;
        woff[j] = n < p.K ? (long long)n * wrow_stride + chunk * 8 : -1;
    }
    // Per-lane SOURCE POINTERS of the k-tile being staged, bumped by 64 bytes per k-tile (0 for lanes that read the
    // zero page) and recomputed only when the filter tap changes: the LOAD half of a phase is then 6 fragment
    // reads + 2 x (m0, global_load_lds) + 2 pointer bumps.  (Deriving the addresses from (r, s, c0) per piece cost
    // ~70 scalar/vector instructions per phase -- longer than the partner wave's 8 MFMAs.)
    const char* xp[2];
    const char* wp[2];
    unsigned xinc[2], winc[2];
    auto retap = [&](int r, int s) {
        const int tp = r * p.S + s;                                          // wave-uniform
        const long long tapdelta = ((long long)(r * p.dh) * p.W + s * p.dw) * p.C;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned bits = tp < 32 ? vlo[j] : vhi[j];
            const bool ok = (bits >> (tp & 31)) & 1u;
            xp[j] = ok ? (const char*)(p.x + tapdelta + xbase[j]) : (const char*)p.zero;
            xinc[j] = ok ? 64u : 0u;
        }
    };
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        wp[j] = woff[j] >= 0 ? (const char*)(p.w + woff[j]) : (const char*)p.zero;
        winc[j] = woff[j] >= 0 ? 64u : 0u;
    }
    retap(0, 0);
    // piece q: 0,1 = x rows, 2,3 = weight rows
    auto stage_piece = [&](int q, int buf) {
        char* xs = smem + buf * STAGE;
        char* ws = xs + BM * ROWB;
        if (q < 2) glds16(xp[q], xs + 16 * (wave + 8 * q) * ROWB);
        else glds16(wp[q - 2], ws + 16 * (wave + 8 * (q - 2)) * ROWB);
    };

    // ---------------- fragment addressing ------------------------------------------------------------
    const int wn = wave & 3;
    const int fr = lane & 31, fh = lane >> 5, swz = (fr >> 2) & 3;
    const int xrow0 = grp * 128, wrow0 = wn * 64;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    unsigned xaddr[2], waddr[2];     // per k16-step of a k-tile: byte address of my 16-byte fragment in stage 0
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const unsigned ko = (unsigned)(((2 * kk + fh) ^ swz) << 4);
        xaddr[kk] = lds0 + (xrow0 + fr) * ROWB + ko;
        waddr[kk] = lds0 + BM * ROWB + (wrow0 + fr) * ROWB + ko;
    }

    // ---------------- epilogue constants + residual prefetch (older than every DMA) --------------------
    const OutT* res = (const OutT*)p.residual;
    ScaleShift8 ss;
    ss.load(p.scale, p.shift, n0 + wrow0 + (lane & 7) * 8, p.K);
    R8<OutT> rres[RESPF ? TM : 1][4];
    if (RESPF && res) {
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int m = m0 + xrow0 + b * 32 + pass * 8 + (lane >> 3);
                const int n = n0 + wrow0 + (lane & 7) * 8;
                const bool ok = m < p.M && n < p.K;
                rres[b][pass].load(res + (ok ? (long long)m * p.K + n : 0));
            }
    }

    f32x16 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    // ---------------- ring prologue: tiles 0 .. 2 ----------------------------------------------------------
    int r = 0, s = 0, c0 = 0;
    auto advance = [&]() {            // to the next k-tile: 32 more channels of this tap, or the next tap
        c0 += 32;
#pragma unroll
        for (int j = 0; j < 2; ++j) wp[j] += winc[j];                        // KRSC rows: taps are contiguous
        if (c0 == p.C) {
            c0 = 0;
            if (++s == p.S) {
                s = 0;
                ++r;
            }
            retap(r, s);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) xp[j] += xinc[j];
        }
    };
    int issued = 0;
#pragma unroll
    for (int i = 0; i < NST - 1; ++i) {
        if (issued < nk) {
#pragma unroll
            for (int q = 0; q < 4; ++q) stage_piece(q, i);
            advance();
            ++issued;
        }
    }
    // tile 0 has to be in LDS before anybody's first fragment read: own pieces (counted), then everybody's
    if (issued >= 3) wait_vm<8>();
    else if (issued == 2) wait_vm<4>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();           // group 1 runs one barrier behind from here on

    // ---------------- main loop: one phase (LOAD half, MFMA half) per 32-wide k-tile -------------------------
    // (a phase per k16-step -- 8 MFMAs -- measured 56% matrix-pipe occupancy on 8192^3: the LOAD half, ~400 cycles
    //  of fragment-read latency, DMA issue and barrier, was longer than the partner's 256 cycles of MFMAs)
    u32x4 af[2][TN], bfm[2][TM];
    auto read_frags = [&](int kk, unsigned sb) {
        lds_read16<0>(af[kk][0], waddr[kk] + sb);
        lds_read16<32 * ROWB>(af[kk][1], waddr[kk] + sb);
        lds_read16<0>(bfm[kk][0], xaddr[kk] + sb);
        lds_read16<32 * ROWB>(bfm[kk][1], xaddr[kk] + sb);
        lds_read16<64 * ROWB>(bfm[kk][2], xaddr[kk] + sb);
        lds_read16<96 * ROWB>(bfm[kk][3], xaddr[kk] + sb);
    };
    auto mma = [&]() {
        // every fragment register is named so that no MFMA can be scheduled above the wait
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(bfm[0][0]), "+v"(bfm[0][1]), "+v"(bfm[0][2]), "+v"(bfm[0][3]),
                       "+v"(af[1][0]), "+v"(af[1][1]), "+v"(bfm[1][0]), "+v"(bfm[1][1]), "+v"(bfm[1][2]), "+v"(bfm[1][3]));
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[kk][a]),
                                                                        __builtin_bit_cast(bf16x8, bfm[kk][b]), acc[a][b],
                                                                        0, 0, 0);
    };
    int cur = 0;
    for (int it = 0; it < nk; ++it) {
        const unsigned sb = (unsigned)(cur * STAGE);
        int rbuf = cur + NST - 1;
        if (rbuf >= NST) rbuf -= NST;
        // ---- LOAD half (at s_setprio(2): its few, latency-critical instructions go first; measured 1150 vs
        //      960 TFLOP/s on 8192^3 against boosting the MFMA half instead) -----------------------------------
        __builtin_amdgcn_s_setprio(2);
        read_frags(0, sb);
        read_frags(1, sb);
        if (issued < nk) {                                // wave-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) stage_piece(q, rbuf);
            advance();
            ++issued;
        }
        // tile it+1 must have landed before the barriers that precede its first read: my pieces of it are older
        // than those of the (at most two) younger tiles in flight
        {
            const int younger = issued - it - 2;          // tiles issued beyond it+1
            if (younger >= 2) wait_vm<8>();
            else if (younger == 1) wait_vm<4>();
            else wait_vm<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my reads have left the stage before the barrier (WAR)
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_s_barrier();
        // ---- MFMA half ----------------------------------------------------------------------------------
        mma();
        __builtin_amdgcn_s_barrier();
        if (++cur == NST) cur = 0;
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();           // balance group 1's extra barrier

    // ---------------- epilogue (as igemm2.hip: LDS transpose, full-line stores) ------------------------------
    __syncthreads();
    char* ep = smem + wave * (32 * EPITCH);
    OutT* y = (OutT*)p.y;
#pragma unroll
    for (int b = 0; b < TM; ++b) {
        R8<OutT> late[4];                 // no prefetch: fetch this pixel tile's 4 residual rows together
        if (!RESPF && res) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int m = m0 + xrow0 + b * 32 + pass * 8 + (lane >> 3);
                const int n = n0 + wrow0 + (lane & 7) * 8;
                const bool ok = m < p.M && n < p.K;
                late[pass].load(res + (ok ? (long long)m * p.K + n : 0));
            }
        }
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = a * 32 + 8 * g + 4 * fh;
                *(float4*)(ep + fr * EPITCH + nl * 4) = make_float4(acc[a][b][4 * g + 0], acc[a][b][4 * g + 1],
                                                                     acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
            }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3), c8 = lane & 7;
            const int m = m0 + xrow0 + b * 32 + row;
            const int n = n0 + wrow0 + c8 * 8;
            const float4 lo = *(const float4*)(ep + row * EPITCH + c8 * 32);
            const float4 hi = *(const float4*)(ep + row * EPITCH + c8 * 32 + 16);
            if (m < p.M && n < p.K) {
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                ss.apply(v);
                if (res) {
                    if constexpr (RESPF) rres[b][pass].add_to(v);
                    else late[pass].add_to(v);
                }
                if (p.act == MV_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (p.act == MV_ACT_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
                }
                long long off = (long long)m * p.K + n;
                if (p.tok > 0) {          // 64 channels = one head = one 128-byte line: same store width, new home
                    const int bi = m / p.tok, ti = m - bi * p.tok;
                    off = (((long long)bi * (p.K >> 6) + (n >> 6)) * p.tok + ti) * 64 + (n & 63);
                }
                Out8<OutT>::st(y + off, v);
            }
        }
    }
}

int igemm3_wanted(long long M, int C, int K, int R, int S) {
    // 256 x 256 tiles: enough of them to fill the chip, a reduction of >= 8 k-tiles of 32, <= 64 filter taps
    const long long ktiles = (long long)R * S * (C / 32);
    const long long tiles = ((M + 255) / 256) * (long long)((K + 255) / 256);
    if (C % 32 != 0 || R * S > 64) return 0;
    if (get_flag("igemm3") == 2) return 1;                // forced (tests): any shape the kernel can express
    (void)tiles;
    return ktiles >= 8;
}

int igemm3_launch(const void* x, const void* w, const float* scale, const float* shift, const void* residual, void* y,
                  int N, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw,
                  int act, int out_dtype, int tok, hipStream_t st) {
    Igemm2P p;
    p.tok = tok;
    p.x2 = nullptr; p.C2 = 0; p.H2 = 0; p.W2 = 0; p.s2 = 1;
    p.dbg = 0;
    p.prof = nullptr;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.residual = residual; p.y = y;
    p.zero = (const bf16_t*)zero_page(st);
    if (!p.zero) {
        set_error("igemm3: zero page allocation failed");
        return MV_E_OOM;
    }
    p.N = N; p.H = H; p.W = W; p.C = C; p.K = K; p.R = R; p.S = S;
    p.Ho = (H + 2 * ph - dh * (R - 1) - 1) / sh + 1;
    p.Wo = (W + 2 * pw - dw * (S - 1) - 1) / sw + 1;
    p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw;
    p.M = (int)((long long)N * p.Ho * p.Wo);
    p.act = act;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (K + 255) / 256;
    const bool dense = (R == 1 && S == 1 && sh == 1 && sw == 1 && ph == 0 && pw == 0);
    constexpr int SMEM = 4 * 512 * 64;
    dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(512);
    set_kernel_name(dense ? "igemm3_bf16_256x256_dense" : "igemm3_bf16_256x256_conv");
#define GO(RP, OT)                                                                                              \
    do {                                                                                                        \
        auto kern = igemm3_kernel<RP, OT>;                                                                      \
        MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));       \
        hipLaunchKernelGGL(kern, grid, block, SMEM, st, p);                                                     \
    } while (0)
    if (out_dtype == MV_F32) GO(false, float);
    else GO(false, bf16_t);        // (the residual prefetch of igemm2 would cost 64 VGPRs here: spills)
#undef GO
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // namespace mv

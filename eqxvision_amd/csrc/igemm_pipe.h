// Shared by the deep-pipelined implicit-GEMM kernels (igemm2.hip, igemm8.hip): launch parameters, inline-asm
// LDS / waitcnt primitives (invisible to hipcc's LDS-DMA alias check) and the residual-row loader.  gfx950 only.
#pragma once
#include "mfma_common.h"

namespace mv {

struct Igemm2P {
    const bf16_t* x;
    const bf16_t* w;
    const float* scale;
    const float* shift;
    const void* residual;
    void* y;
    const bf16_t* zero;
    int N, H, W, C, K, R, S, Ho, Wo, sh, sw, ph, pw, dh, dw;
    int M, tiles_m, tiles_n, act;   // M = rows covered by THIS launch (rows 0 .. M-1)
    int dbg;                        // experiments only
    long long* prof;                // experiments only: per-block phase stamps
    const bf16_t* x2;               // igemm2 DUAL: second reduction source, NHWC [N][H2][W2][C2], read at pixel stride s2
    int C2, H2, W2, s2;
    int gm;                         // igemm8: pixel tiles per group of the tile order (mfma_common.h: tile_coords)
    int skew;                       // igemm8 experiment: start delay of the first-round workgroups, (block / 8 & 3) x skew x 10 ns
    int tok;                        // > 0: head-major output y[b][n/64][t][n%64], rows m = b*tok + t (qkv projection)
    unsigned* sync;                 // igemm8s split-K: two words per tile (arrivals, partial-is-there), zero between launches; nullptr = no split
    float* ws;                      // igemm8s split-K: one fp32 tile of partial sums per tile
    // LayerNorm folded across two Linears (igemm8, 256 x 256 tiles only; see epilogue_rows):
    void* y2;                       // producer: low plane of the rows y (y = the high plane = the next Linear's operand)
    const void* residual2;          // producer: low plane of the residual rows (residual = their high plane); nullptr: residual is fp32 rows
    float* stats_out;               // producer: [ceil(K / 256)][M][2] = per row and 256-column tile (sum, sum of squares about the tile's own mean)
    const float* stats_in;          // consumer: the same table for THIS launch's operand rows, [ceil(C / 256)][M][2]
    float ln_eps;                   // consumer
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int IMM> __device__ __forceinline__ void lds_read16(u32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename OutT> struct R8;
template <> struct R8<bf16_t> {
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
template <> struct R8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
    __device__ __forceinline__ void add_to(float* v) const {
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Branch-free row-major read-back (round 5).  The first version of this epilogue wrapped every residual load and every store
// in `if (m < M && n < K)`: hipcc turns that into an exec-masked region per pass, and because gfx9's vmcnt counts loads AND
// stores in issue order and the waits for the (long finished) scale / shift and residual loads sat INSIDE those regions, its
// wait-count pass fell back to `s_waitcnt vmcnt(0)` in every pass -- 16 serialised store round trips per wave and tile
// (3.7 us of a 5.2 us bf16 epilogue, profiles/r04/vit_gemm_tile_phases.txt "no stores").  Here rows and columns outside the
// tensor get an out-of-range offset into a raw buffer descriptor instead (the buffer unit drops the store / returns zeros), so
// the epilogue has no divergent control flow, every wait is an exact count and the stores of a tile stream behind each other.

template <typename OutT> struct Buf8;      // 8 consecutive channels of one row, through a buffer descriptor
template <> struct Buf8<bf16_t> {
    u32x4 u;
    __device__ __forceinline__ void load(brsrc_t r, unsigned vo, unsigned so) { u = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0); }
    __device__ __forceinline__ void add_to(float* v) const {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] += __uint_as_float(u[e] << 16);
            v[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store(brsrc_t r, unsigned vo, const float* v) {
        u32x4 o;
        o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]); o[2] = pack_bf2(v[4], v[5]); o[3] = pack_bf2(v[6], v[7]);
        __builtin_amdgcn_raw_buffer_store_b128(o, r, vo, 0, 0);
    }
};
template <> struct Buf8<float> {
    u32x4 a, b;
    __device__ __forceinline__ void load(brsrc_t r, unsigned vo, unsigned so) {
        a = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
        b = __builtin_amdgcn_raw_buffer_load_b128(r, vo + 16u, so, 0);
    }
    __device__ __forceinline__ void add_to(float* v) const {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(a[e]); v[4 + e] += __uint_as_float(b[e]); }
    }
    static __device__ __forceinline__ void store(brsrc_t r, unsigned vo, const float* v) {
        u32x4 lo, hi;
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = __float_as_uint(v[e]); hi[e] = __float_as_uint(v[4 + e]); }
        __builtin_amdgcn_raw_buffer_store_b128(lo, r, vo, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(hi, r, vo + 16u, 0, 0);
    }
};

// One wave writes its (NBT x 32 pixel rows) x 64 channels: accumulator tiles acc[a][b] (a = channel half, b = pixel tile) go
// through the wave-private LDS patch `ep` (32 rows of EPITCH bytes) and leave as full 128-byte (bf16) / 256-byte (fp32) lines.
//   mrow0 = first pixel row of the wave, ncol0 = first channel of the wave (multiple of 64), both wave-uniform.
//
// LNF 2: the consumer side of a LayerNorm folded across two Linears (epilogue_rows_ln below):
//   v = rstd[row] * acc + (-mean * rstd)[row] * colsum[n] + b'[n]; `lnst` = the wave group's 128 x (-mean * rstd, rstd) table in LDS
//   (finalised by the kernel from the producers' pieces), ss.sc = colsum, ss.sh = b'.
__device__ __forceinline__ float dpp_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_half_mirror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true)); }
__device__ __forceinline__ float sum8_lanes(float v) {       // sum over the 8 consecutive lanes (lane & ~7) .. (lane | 7), in every lane
    v += dpp_xor1(v);
    v += dpp_xor2(v);
    v += dpp_half_mirror(v);
    return v;
}

template <typename OutT, bool LIN, int NBT, int EPITCH, int LNF = 0>
__device__ __forceinline__ void epilogue_rows(const Igemm2P& p, char* ep, f32x16 (&acc)[2][NBT], const ScaleShift8& ss,
                                              const OutT* res, bool do_store, int mrow0, int ncol0, int lane,
                                              const char* lnst = nullptr) {
    static_assert(LNF == 0 || (LNF == 2 && sizeof(OutT) == 2 && !LIN), "LayerNorm fold, consumer form");
    constexpr unsigned SZ = sizeof(OutT);
    const int fr = lane & 31, fh = lane >> 5, r0 = lane >> 3, c8 = lane & 7;
    const int rl = ncol0 + c8 * 8 < p.K ? p.M - mrow0 : 0;           // rows of this wave's strip the lane may touch (0: its channels are past K)
    const long long wave_elem = (long long)mrow0 * p.K + ncol0;
    const bool has_res = res != nullptr;
    const brsrc_t rr = make_brsrc(has_res ? res + wave_elem : (const OutT*)p.y, has_res);
    const unsigned vrow = (unsigned)(r0 * p.K + c8 * 8) * SZ;          // lane's element of the wave's first 8 rows
    const unsigned kstep = 8u * (unsigned)p.K * SZ;                    // 8 rows further down
    // head-major token layout (tok > 0): y[b][n / 64][t][n % 64], rows m = b * tok + t
    const bool hm = p.tok > 0;
    int bi0 = 0, ti0 = 0;
    if (hm) { bi0 = mrow0 / p.tok; ti0 = mrow0 - bi0 * p.tok; }
    const long long y_elem = hm ? ((long long)bi0 * (p.K >> 6) + (ncol0 >> 6)) * p.tok * 64 : wave_elem;
    const brsrc_t ry = make_brsrc((OutT*)p.y + y_elem, do_store);
    const float inv_tok = hm ? 1.0f / (float)p.tok : 0.f;
    const unsigned img_step = (unsigned)(p.K >> 6) * (unsigned)p.tok * 64u;   // elements from image b to b + 1 (same channel block)
    // Residual loads: row-major offset in a VGPR (or out of range), the 8-row step in the scalar offset.
    // Stores: the WHOLE offset in the VGPR and soffset = 0.  A buffer store of more than 64 bits reads its data registers over
    // several cycles; hipcc's hazard recogniser only inserts the wait state before a VALU overwrite of those registers when the
    // store's soffset is NOT a register (GCNHazardRecognizer: "this hazard only exists if the instruction is not using a register
    // in the soffset field") -- on gfx950 the overwrite corrupted one dword of the second fp32 store of a pass in some lanes
    // (tools/dbg_epi.py: rows 25/27/29/31 of a strip, first channel of the upper half) when soffset was an SGPR.
    auto row_vo = [&](int row) -> unsigned { return row + r0 < rl ? vrow : BUF_OOB; };
    auto y_vo = [&](int row, int step) -> unsigned {
        unsigned rel = vrow + (unsigned)step * kstep;
        if (hm) {
            const int v = ti0 + row + r0;                            // < tok + 128
            int q = (int)((float)v * inv_tok), t = v - q * p.tok;
            q += t >= p.tok ? 1 : (t < 0 ? -1 : 0);
            t -= t >= p.tok ? p.tok : (t < 0 ? -p.tok : 0);
            rel = ((unsigned)q * img_step + (unsigned)t * 64u + (unsigned)c8 * 8u) * SZ;
        }
        return row + r0 < rl ? rel : BUF_OOB;
    };
    Buf8<OutT> late[2][4];
    auto fetch_res = [&](int b, Buf8<OutT>(&dst)[4]) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) dst[pass].load(rr, row_vo(b * 32 + pass * 8), (unsigned)(b * 4 + pass) * kstep);
    };
    // (Fetching the first tile's rows BEFORE the main loop -- 32 more live registers, 255 VGPRs in the lin / fp32 kernel, no spills --
    // was built in round 6: proj 61.4 -> 62.2 us, fc2 160.0 -> 157.4 us alone, vit_base 22 784 vs 22 805 img/s: nothing.
    // profiles/r06/vit_residual_rows_before_main_loop_ab.txt)
    fetch_res(0, late[0]);
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
        if (b + 1 < NBT) fetch_res(b + 1, late[(b + 1) & 1]);       // one pixel tile ahead: its latency hides under this tile's stores
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = a * 32 + 8 * g + 4 * fh;
                *(float4*)(ep + fr * EPITCH + nl * 4) = make_float4(acc[a][b][4 * g + 0], acc[a][b][4 * g + 1],
                                                                     acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
            }
        wave_lds_fence();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + r0;
            const float4 lo = *(const float4*)(ep + row * EPITCH + c8 * 32);
            const float4 hi = *(const float4*)(ep + row * EPITCH + c8 * 32 + 16);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if constexpr (LNF == 2) {
                const float2 st = *(const float2*)(lnst + (b * 32 + row) * 8);          // (-mean * rstd, rstd) of this row
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(st.y, v[e], fmaf(st.x, ss.sc[e], ss.sh[e]));
            } else if constexpr (LIN) ss.apply_shift(v);
            else ss.apply(v);
            if (has_res) late[b & 1][pass].add_to(v);
            const unsigned voy = y_vo(b * 32 + pass * 8, b * 4 + pass);
            if (p.act == MV_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if constexpr (sizeof(OutT) == 2) {
                u32x4 o;
                if (p.act == MV_ACT_GELU_TANH) {     // GELU on packed fp32 pairs (v_pk_fma / v_pk_mul), straight to the bf16 words
                    o[0] = gelu_tanh_pack2(v[0], v[1]); o[1] = gelu_tanh_pack2(v[2], v[3]);
                    o[2] = gelu_tanh_pack2(v[4], v[5]); o[3] = gelu_tanh_pack2(v[6], v[7]);
                } else {
                    o[0] = pack_bf2(v[0], v[1]); o[1] = pack_bf2(v[2], v[3]); o[2] = pack_bf2(v[4], v[5]); o[3] = pack_bf2(v[6], v[7]);
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, ry, voy, 0, 0);
            } else {
                if (p.act == MV_ACT_GELU_TANH) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
                }
                Buf8<OutT>::store(ry, voy, v);
            }
        }
        wave_lds_fence();
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// The LayerNorm between two Linears folded into their epilogues (ViT: norm1 -> qkv, norm2 -> fc1; vit.py:139-157):
//   LN(y) . W^T + b  =  rstd * (y . W'^T - mean * colsum(W')) + b',   W' = W . diag(gamma),  b' = b + W . beta
// so the consumer's MAIN LOOP runs on the un-normalised rows (rounded to bf16 once) and only its epilogue needs the row statistics.
// This is the PRODUCER: y = residual + acc + shift.  The residual stream is kept as TWO bf16 planes, hi = bf16(y) and
// lo = bf16(y - hi) (the same four bytes per value as fp32, ~16 mantissa bits: 2^-17 relative per store against the 2^-9 of the
// operands) -- the high plane IS the consumer's operand, so folding the LayerNorm costs no extra copy of the rows:
//   RES_SPLIT  the residual rows arrive as planes (p.residual, p.residual2), else as fp32 rows (the first block: the token rows);
//   OUT_SPLIT  y leaves as planes (p.y, p.y2), and per row and 64-column piece -- a wave's 64 channels -- the pair (sum, sum of squared
//              deviations from the piece's own mean) goes to `slots` (LDS: [row of the wave group][4 waves] float2; the 8 lanes of a
//              row meet through three DPP adds); the kernel merges the four waves of a row in a fixed order behind a barrier
//              (merge_row_stats) and stores ONE pair per row and 256-column tile.  Chan's parallel form throughout: no
//              E[x^2] - mean^2 cancellation whatever the row's offset is.  Else y leaves as fp32 rows (the last block).
//   `slots` = the wave group's first row, `wc` = the wave's 64-column piece of the tile, `dummy` = 512 bytes per wave nobody reads.
template <bool RES_SPLIT, bool OUT_SPLIT, int NBT, int EPITCH>
__device__ __forceinline__ void epilogue_rows_ln(const Igemm2P& p, char* ep, f32x16 (&acc)[2][NBT], const ScaleShift8& ss, bool do_store,
                                                 int mrow0, int ncol0, int lane, char* slots, int wc, char* dummy) {
    static_assert(RES_SPLIT || OUT_SPLIT, "fp32 in, fp32 out is the plain epilogue");
    const int fr = lane & 31, fh = lane >> 5, r0 = lane >> 3, c8 = lane & 7;
    const int rl = ncol0 + c8 * 8 < p.K ? p.M - mrow0 : 0;
    const long long wave_elem = (long long)mrow0 * p.K + ncol0;
    const unsigned vrow2 = (unsigned)(r0 * p.K + c8 * 8) * 2u, kstep2 = 16u * (unsigned)p.K;       // bf16 planes
    const brsrc_t rra = make_brsrc(RES_SPLIT ? (const void*)((const bf16_t*)p.residual + wave_elem) : (const void*)((const float*)p.residual + wave_elem));
    const brsrc_t rrb = make_brsrc(RES_SPLIT ? (const void*)((const bf16_t*)p.residual2 + wave_elem) : (const void*)p.y, RES_SPLIT);
    const brsrc_t rya = make_brsrc(OUT_SPLIT ? (void*)((bf16_t*)p.y + wave_elem) : (void*)((float*)p.y + wave_elem), do_store);
    const brsrc_t ryb = make_brsrc(OUT_SPLIT ? (void*)((bf16_t*)p.y2 + wave_elem) : p.y, OUT_SPLIT && do_store);
    auto row_ok = [&](int row) -> bool { return row + r0 < rl; };
    u32x4 la[2][4], lb[2][4];
    auto fetch_res = [&](int b, u32x4 (&da)[4], u32x4 (&db)[4]) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const bool ok = row_ok(b * 32 + pass * 8);
            const unsigned step = (unsigned)(b * 4 + pass);
            if constexpr (RES_SPLIT) {
                da[pass] = __builtin_amdgcn_raw_buffer_load_b128(rra, ok ? vrow2 : BUF_OOB, step * kstep2, 0);
                db[pass] = __builtin_amdgcn_raw_buffer_load_b128(rrb, ok ? vrow2 : BUF_OOB, step * kstep2, 0);
            } else {
                da[pass] = __builtin_amdgcn_raw_buffer_load_b128(rra, ok ? 2u * vrow2 : BUF_OOB, step * 2u * kstep2, 0);
                db[pass] = __builtin_amdgcn_raw_buffer_load_b128(rra, ok ? 2u * vrow2 + 16u : BUF_OOB, step * 2u * kstep2, 0);
            }
        }
    };
    fetch_res(0, la[0], lb[0]);
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
        if (b + 1 < NBT) fetch_res(b + 1, la[(b + 1) & 1], lb[(b + 1) & 1]);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = a * 32 + 8 * g + 4 * fh;
                *(float4*)(ep + fr * EPITCH + nl * 4) = make_float4(acc[a][b][4 * g + 0], acc[a][b][4 * g + 1],
                                                                     acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
            }
        wave_lds_fence();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + r0;
            const float4 lo = *(const float4*)(ep + row * EPITCH + c8 * 32);
            const float4 hi = *(const float4*)(ep + row * EPITCH + c8 * 32 + 16);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            ss.apply_shift(v);
            const u32x4 ra = la[b & 1][pass], rb = lb[b & 1][pass];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (RES_SPLIT) {
                    v[2 * e] += __uint_as_float(ra[e] << 16) + __uint_as_float(rb[e] << 16);
                    v[2 * e + 1] += __uint_as_float(ra[e] & 0xffff0000u) + __uint_as_float(rb[e] & 0xffff0000u);
                } else {
                    v[e] += __uint_as_float(ra[e]);
                    v[4 + e] += __uint_as_float(rb[e]);
                }
            }
            const bool ok = row_ok(b * 32 + pass * 8);
            const unsigned step = (unsigned)(b * 4 + pass);
            if constexpr (OUT_SPLIT) {
                const float s = sum8_lanes(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
                const float mu = s * (1.0f / 64.0f);
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[e] - mu; q = fmaf(d, d, q); }
                q = sum8_lanes(q);
                // lane c8 == 0 of a row files the pair; the other seven write it to the wave's dummy words (no divergent region)
                *(float2*)(c8 == 0 ? slots + ((b * 32 + row) * 4 + wc) * 8 : dummy + lane * 8) = make_float2(s, q);
                u32x4 oh, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    oh[e] = pack_bf2(v[2 * e], v[2 * e + 1]);
                    ol[e] = pack_bf2(v[2 * e] - __uint_as_float(oh[e] << 16), v[2 * e + 1] - __uint_as_float(oh[e] & 0xffff0000u));
                }
                const unsigned vo = ok ? vrow2 + step * kstep2 : BUF_OOB;       // whole offset in the VGPR (see epilogue_rows)
                __builtin_amdgcn_raw_buffer_store_b128(oh, rya, vo, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(ol, ryb, vo, 0, 0);
            } else {
                Buf8<float>::store(rya, ok ? 2u * (vrow2 + step * kstep2) : BUF_OOB, v);
            }
        }
        wave_lds_fence();
    }
}


// One thread per row of the 256-row tile: the four waves' (sum, m2) of the row -> one pair for the tile's `nw` valid 64-column pieces
// (Chan's merge in a fixed order), stored to stats_out[tile_n][m0 + row].
__device__ __forceinline__ void merge_row_stats(const Igemm2P& p, const char* slots, int tid, int m0, int tile_n, int nw) {
    const float4 a = *(const float4*)(slots + tid * 32), b = *(const float4*)(slots + tid * 32 + 16);
    const float s[4] = {a.x, a.z, b.x, b.z}, q[4] = {a.y, a.w, b.y, b.w};
    float tot = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) tot += j < nw ? s[j] : 0.f;
    const float mean = tot / (float)(64 * nw);
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d = s[j] * (1.0f / 64.0f) - mean;
        m2 += j < nw ? q[j] + 64.0f * d * d : 0.f;
    }
    const brsrc_t rst = make_brsrc(p.stats_out + 2 * ((long long)tile_n * p.M + m0));
    u32x2 st;
    st[0] = __float_as_uint(tot); st[1] = __float_as_uint(m2);
    __builtin_amdgcn_raw_buffer_store_b64(st, rst, m0 + tid < p.M ? 8u * (unsigned)tid : BUF_OOB, 0, 0);
}

}  // namespace mv

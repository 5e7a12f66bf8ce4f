// The attention half of a Swin block in ONE launch, one workgroup per window (gfx950):
//     y = x + proj(window_attention(qkv(LayerNorm(x))))          (swin.py:572-578 first line, 90-255, 342-366)
// for heads of 32 channels and 7 x 7 windows: C = 384 (stage 2 of swin_t / swin_s: 12 heads, one window per workgroup), C = 192
// (stage 1: 6 heads, TWO windows per workgroup) and C = 96 (stage 0: 3 heads, FOUR windows per workgroup) -- windows x heads-per-
// group = 4 in every case, so the same 24-MFMA-tile GEMM phases and the same 8 attention jobs fill the 8 waves.  The un-fused path is four launches (LayerNorm, qkv Linear, window attention, proj Linear +
// residual) that move the 1152-channel qkv tensor and the attention output through HBM; here a window's 49 token rows are
// gathered once (cyclic shift + window partition = index arithmetic), and q / k / v / the attention output live in LDS:
//
//   phase 0   LayerNorm of the 49 rows (fp32 statistics) -> bf16 rows in LDS (rows 49..63 zero; LayerNorm affine folded into
//             the qkv weights by the host)
//   3 x       heads in groups of HG = 4 / windows: [q | k | v] of the group = 3 HG channel tiles x (2 x windows) token blocks = 24 MFMA tiles, three per
//             wave (A = weights straight from L2 in fragment order, B = token rows from LDS, as ln_mlp_stream.hip);
//             q, k go to LDS token-major, v TRANSPOSED element by element in the key order the P.V product wants;
//             then attention, wave = (head of the group, query block of 32): S^T = K.Q^T puts one query per lane, scale +
//             relative-position bias + shift mask (-100, as the reference) + softmax in registers (one lane^32 exchange),
//             P feeds the P.V MFMA from the accumulator registers it already sits in; output -> LDS, token-major
//   proj      the same three-tiles-per-wave GEMM over the attention output; + bias -> fp32 tile in LDS
//   epilogue  whole rows: + residual row (coalesced fp32 re-read), stored back to the token's own position (window reverse +
//             roll back = the position it was gathered from)
#include <type_traits>

#include "mfma_common.h"

namespace mv {

namespace {

struct SwinBAP {
    const float* x;        // [B][Hf][Wf][C] fp32 residual stream
    const bf16_t* wqkv;    // [3 groups][3 HG tiles: q heads, k heads, v heads of the group][C/16][64][8]
    const float* bqkv;     // [groups][3 HG][32]
    const bf16_t* wp;      // [C/32 tiles][C/16][64][8]
    const float* bp;       // [C]
    const float* bias;     // [heads][64][64]: relative-position bias, -1e30 on padded keys
    float* y;
    int Hf, Wf, shh, shw, nWw, nW;
    float eps;
    int ablate;            // debug build only (w96_ablate): 1 no weight streaming, 2 no bias rows, 4 no stores, 8 no gather
    long long* prof;       // experiments only (tools/time_swin_block_attn.py): per-wave wall-clock stamps at the phase boundaries
};

// NWV = waves per workgroup: 8, or 4 for C = 96 (two windows, one head at a time: half the LDS, so that TWO workgroups share a CU --
// a stage-0 workgroup has 1.5 us of matrix work and ~22 us of dependent latencies: gather, 7 barriers, bias loads, epilogue)
template <int C, int NWIN, int NWV>
__global__ __launch_bounds__(NWV * 64) void swin_block_attn_kernel(const SwinBAP p) {
    constexpr int NT = NWV * 64, NSLOT = NWV / 2;            // threads; (head of group, window) slots = attention jobs / 2
    constexpr int NH = C / 32, HG = NSLOT / NWIN, NG = NH / HG, KS = C / 16, NTOK = 49, WS = 7;
    // k-loop: unrolled by U; weight fragments D steps ahead, token fragments 2 steps ahead in a ring of PB (all static indices)
    // (8 / 6 fragments in flight per stream at C = 384: 49.7 us -- 256 VGPRs -- / 39.2 us against 39.7: not the limit)
    constexpr int U = (KS % 4 == 0) ? 4 : 6, D = (KS % 4 == 0) ? 4 : 3, PB = (KS % 4 == 0) ? 4 : 3;
    constexpr int NTB = 2 * NWIN, TR = 64 * NWIN, GT = 3 * HG, NCT = C / 32;      // token blocks / rows per workgroup; tiles per group
    static_assert(NWIN * HG == NSLOT && NH % HG == 0 && C % 48 == 0 && KS % U == 0 && U % D == 0 && U % PB == 0 && GT * NTB == 3 * NWV &&
                  NCT * NTB == 3 * NWV, "layout");
    // token rows of the normalised input / the attention output: 2 C bytes + 16 of padding (conflict-free 16-byte fragment reads);
    // C = 96: the padding would not fit -- 16 bytes after every FOURTH row do the same job (192-byte rows start 48 dwords apart)
    constexpr int XBYTES = C == 96 ? TR * 192 + (TR / 4) * 16 : TR * (C * 2 + 16);
    auto xoff = [](int r) -> int { return C == 96 ? r * 192 + (r >> 2) * 16 : r * (C * 2 + 16); };
    constexpr int QROW = 80, VROW = 144;                 // q, k rows: 32 dh (+ pad); v^T rows: 64 keys (+ pad)
    constexpr int LDS_NX = 0;
    constexpr int LDS_Q = XBYTES;                        // [NSLOT (head of group, window)][64 tokens][QROW]
    constexpr int LDS_K = LDS_Q + NSLOT * 64 * QROW;
    constexpr int LDS_V = LDS_K + NSLOT * 64 * QROW;     // [NSLOT][32 dh][VROW]
    constexpr int LDS_O = LDS_V + NSLOT * 32 * VROW;
    constexpr int LDS_REG = LDS_O + XBYTES;              // int[NWIN][64]: shift-mask region per token
    constexpr int YROW = C * 4 + 16;
    static_assert(TR * YROW <= LDS_O && LDS_REG + NWIN * 256 <= 160 * 1024, "result tile must fit below the attention output");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int wgs = p.nW / NWIN;                          // workgroups per image
    const int b = blockIdx.x / wgs, wloc0 = (blockIdx.x - b * wgs) * NWIN;
    long long st_w[8];
    int nst = 0;
#define MV_SBA_STAMP() do { if (p.prof && nst < 8) st_w[nst++] = wall_clock64(); } while (0)
    MV_SBA_STAMP();                                       // 0: start
    const bool shifted = (p.shh + p.shw) > 0;
    auto tok_row = [&](int r) -> long long {            // row r = 64 * window + token -> its row in x / y (un-rolled position)
        const int wloc = wloc0 + (r >> 6), t = r & 63;
        const int wy = wloc / p.nWw, wx = wloc - wy * p.nWw;
        const int ty = t / WS, tx = t - ty * WS;
        int oy = wy * WS + ty + p.shh, ox = wx * WS + tx + p.shw;
        if (oy >= p.Hf) oy -= p.Hf;
        if (ox >= p.Wf) ox -= p.Wf;
        return ((long long)b * p.Hf + oy) * p.Wf + ox;
    };

    // ---------------- weight-fragment streams: units 3w .. 3w+2 of (tiles x NTB token blocks) touch at most two channel tiles ------
    const int ta = (3 * wave) / NTB;                                   // first tile; the second stream is tile ta + 1 (clamped)
    const int u_tile[3] = {(3 * wave) / NTB, (3 * wave + 1) / NTB, (3 * wave + 2) / NTB};
    const int u_tb[3] = {(3 * wave) % NTB, (3 * wave + 1) % NTB, (3 * wave + 2) % NTB};
    const int pat = (u_tile[1] - ta) + 2 * (u_tile[2] - ta);           // which stream each unit multiplies with: 0 = 000, 2 = 001, 3 = 011
    auto tile_base = [&](const bf16_t* w, int tile, int ntiles) -> const uint4* {
        tile = tile < ntiles ? tile : ntiles - 1;
        return (const uint4*)w + (size_t)tile * KS * 64 + lane;
    };
    uint4 a0[D], a1[D];
    {
        const uint4* s0 = tile_base(p.wqkv, ta, GT);
        const uint4* s1 = tile_base(p.wqkv, ta + 1, GT);
#pragma unroll
        for (int d = 0; d < D; ++d) { a0[d] = s0[d * 64]; a1[d] = s1[d * 64]; }
    }

    // ---------------- phase 0: gather + LayerNorm -> LDS ----------------------------------------------------------------------
    {
        constexpr int LPR = C / 12, RPP = 64 / LPR, RPW = TR / NWV, NP = RPW / RPP;      // lanes per row (3 float4 each), rows per pass / wave
        static_assert(NP * RPP == RPW, "whole passes");
        const int lr = lane / LPR, lq = lane % LPR;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int r = RPW * wave + ps * RPP + lr;
            const bool real = (r & 63) < NTOK;
            const float4* src = (const float4*)(p.x + tok_row(real ? r : (r & ~63)) * C);
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
            const float rstd = real ? rsqrtf(q * (1.0f / C) + p.eps) : 0.f;            // padded rows: zeros
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                uint2 u;
                u.x = pack_bf2(v[i].x * rstd, v[i].y * rstd);
                u.y = pack_bf2(v[i].z * rstd, v[i].w * rstd);
                *(uint2*)(smem + LDS_NX + xoff(r) + (lq + LPR * i) * 8) = u;
            }
        }
        if (tid < TR) {                                  // shift-mask region of every token (rolled coordinates, swin.py:190-209)
            const int wloc = wloc0 + (tid >> 6);
            const int wy = wloc / p.nWw, wx = wloc - wy * p.nWw;
            const int t = (tid & 63) < NTOK ? (tid & 63) : NTOK - 1;
            const int ty = t / WS, tx = t - ty * WS;
            const int yy = wy * WS + ty, xx = wx * WS + tx;
            const int rh = (yy < p.Hf - WS) ? 0 : (yy < p.Hf - p.shh ? 1 : 2);
            const int rw = (xx < p.Wf - WS) ? 0 : (xx < p.Wf - p.shw ? 1 : 2);
            ((int*)(smem + LDS_REG))[tid] = shifted ? rh * 3 + rw : 0;
        }
    }
    __syncthreads();
    MV_SBA_STAMP();                                       // 1: LayerNorm + barrier

    // ---------------- the three-units-per-wave GEMM: acc[i] = unit i of this wave over K = C ----------------------------------------
    f32x16 acc[3];
    // src: LDS byte offset of the token rows (B operand); cur0 / cur1: this phase's two fragment streams; nx0 / nx1: the next phase's.
    // PAT: which stream each unit multiplies with (bit i-1 of PAT = unit i uses the second stream) -- a template value, chosen once
    // per wave: no selects or branches inside the loop.
    auto gemm3 = [&](auto patc, int src, const uint4* cur0, const uint4* cur1, const uint4* nx0, const uint4* nx1)
        __attribute__((always_inline)) {        // (not inlined, the captured accumulators and fragment rings live in scratch: 18x slower)
        constexpr int PAT = decltype(patc)::value;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        const char* xb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) xb[i] = smem + src + xoff(32 * u_tb[i] + fr) + fh * 16;
        bf16x8 bq[PB][3];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 3; ++i) bq[t][i] = *(const bf16x8*)(xb[i] + t * 32);
        for (int j0 = 0; j0 < KS; j0 += U) {
#pragma unroll
            for (int d = 0; d < U; ++d) {
                const int j = j0 + d;
                const int jn = j + 2 < KS ? j + 2 : KS - 1;
                const bool in = j + D < KS;
                const uint4* n0 = in ? cur0 + (size_t)(j + D) * 64 : nx0 + (size_t)(j + D - KS) * 64;
                const uint4* n1 = in ? cur1 + (size_t)(j + D) * 64 : nx1 + (size_t)(j + D - KS) * 64;
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 f0 = __builtin_bit_cast(bf16x8, a0[d % D]);
                const bf16x8 f1 = __builtin_bit_cast(bf16x8, a1[d % D]);
                a0[d % D] = *n0;
                a1[d % D] = *n1;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, bq[d % PB][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((PAT & 1) ? f1 : f0, bq[d % PB][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((PAT & 2) ? f1 : f0, bq[d % PB][2], acc[2], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 3; ++i) bq[(d + 2) % PB][i] = *(const bf16x8*)(xb[i] + jn * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto gemm = [&](int src, const uint4* cur0, const uint4* cur1, const uint4* nx0, const uint4* nx1) __attribute__((always_inline)) {
        if (pat == 0) gemm3(std::integral_constant<int, 0>{}, src, cur0, cur1, nx0, nx1);
        else if (pat == 2) gemm3(std::integral_constant<int, 2>{}, src, cur0, cur1, nx0, nx1);
        else gemm3(std::integral_constant<int, 3>{}, src, cur0, cur1, nx0, nx1);
    };

    const int* regl = (const int*)(smem + LDS_REG);
    const float scale = rsqrtf(32.0f);

    for (int g = 0; g < NG; ++g) {
        // ---------------- q | k | v of heads HG g .. HG g + HG - 1, all windows of the workgroup -----------------------------------------
        const bf16_t* wg = p.wqkv + (size_t)g * GT * KS * 64 * 8;
        const bool last = g + 1 == NG;
        const bf16_t* wn = last ? p.wp : wg + (size_t)GT * KS * 64 * 8;                 // next phase: next group, then proj
        const int ntn = last ? NCT : GT;
        gemm(LDS_NX, tile_base(wg, ta, GT), tile_base(wg, ta + 1, GT), tile_base(wn, ta, ntn), tile_base(wn, ta + 1, ntn));
        if (g == 0) MV_SBA_STAMP();                       // 2: qkv GEMM of group 0
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int tile = u_tile[i], tb = u_tb[i];
            const int kind = tile / HG, hl = tile - kind * HG;          // 0 q, 1 k, 2 v; head of the group
            const int win = tb >> 1, slotq = hl * NWIN + win;           // (head of group, window) slot of the q / k / v buffers
            const float* bb = p.bqkv + ((size_t)g * GT + tile) * 32 + 4 * fh;
            const int tok = 32 * (tb & 1) + fr;
            if (kind < 2) {
                char* dst = smem + (kind == 0 ? LDS_Q : LDS_K) + (slotq * 64 + tok) * QROW + 8 * fh;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 bv = *(const float4*)(bb + 8 * gq);
                    uint2 u;
                    u.x = pack_bf2(acc[i][4 * gq + 0] + bv.x, acc[i][4 * gq + 1] + bv.y);
                    u.y = pack_bf2(acc[i][4 * gq + 2] + bv.z, acc[i][4 * gq + 3] + bv.w);
                    *(uint2*)(dst + 16 * gq) = u;
                }
            } else {
                // v^T[dh][slot(key)]: the 8 keys a lane feeds to one P.V step sit next to each other (slot order (token block, pair
                // of groups, lane half, group parity, r)) -- the lane's own position decides the slot of its token
                const int kg = fr >> 3, khh = (fr >> 2) & 1, kr = fr & 3;
                const int slot = (((tb & 1) * 2 + (kg >> 1)) * 2 + khh) * 8 + (kg & 1) * 4 + kr;
                char* dst = smem + LDS_V + slotq * 32 * VROW + slot * 2;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 bv = *(const float4*)(bb + 8 * gq);
                    const float vv[4] = {acc[i][4 * gq + 0] + bv.x, acc[i][4 * gq + 1] + bv.y, acc[i][4 * gq + 2] + bv.z,
                                         acc[i][4 * gq + 3] + bv.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) *(bf16_t*)(dst + (8 * gq + 4 * fh + e) * VROW) = f2bf(vv[e]);
                }
            }
        }
        __syncthreads();
        if (g == 0) MV_SBA_STAMP();                       // 3: q / k / v stores + barrier
        // ---------------- attention: wave = ((head of the group, window), query block qb) ------------------------------------------------
        {
            const int slotq = wave >> 1, qb = wave & 1;
            const int hl = slotq / NWIN, win = slotq - hl * NWIN, h = HG * g + hl;
            const int qtok = 32 * qb + fr;
            const char* qp = smem + LDS_Q + (slotq * 64 + qtok) * QROW + fh * 16;
            const char* kp = smem + LDS_K + (slotq * 64 + fr) * QROW + fh * 16;
            bf16x8 qf[2], kf[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                qf[j] = *(const bf16x8*)(qp + j * 32);
                kf[0][j] = *(const bf16x8*)(kp + j * 32);
                kf[1][j] = *(const bf16x8*)(kp + 32 * QROW + j * 32);
            }
            f32x16 s[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kt][j], qf[j], s[kt], 0, 0, 0);
            }
            const float* brow = p.bias + ((size_t)h * 64 + qtok) * 64;
            const int* rl = regl + 64 * win;
            const int qreg = rl[qtok];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int key0 = 32 * kt + 8 * gq + 4 * fh;
                    const float4 bv = *(const float4*)(brow + key0);
                    const int4 kr = *(const int4*)(rl + key0);
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                    const int rr[4] = {kr.x, kr.y, kr.z, kr.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = fmaf(s[kt][4 * gq + e], scale, bb[e]);
                        if (rr[e] != qreg) v += -100.0f;
                        s[kt][4 * gq + e] = v;
                        mx = fmaxf(mx, v);
                    }
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pe = __expf(s[kt][e] - mx);
                    s[kt][e] = pe;
                    sum += pe;
                }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.f / sum;
            f32x16 o;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = 0.f;
            const char* vp = smem + LDS_V + (slotq * 32 + fr) * VROW + fh * 16;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    uint4 pf;
                    pf.x = pack_bf2(s[kt][8 * gp + 0], s[kt][8 * gp + 1]);
                    pf.y = pack_bf2(s[kt][8 * gp + 2], s[kt][8 * gp + 3]);
                    pf.z = pack_bf2(s[kt][8 * gp + 4], s[kt][8 * gp + 5]);
                    pf.w = pack_bf2(s[kt][8 * gp + 6], s[kt][8 * gp + 7]);
                    const bf16x8 vf = *(const bf16x8*)(vp + (kt * 2 + gp) * 32);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pf), o, 0, 0, 0);
                }
            char* od = smem + LDS_O + xoff(64 * win + qtok) + (32 * h + 4 * fh) * 2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 u;
                u.x = pack_bf2(o[4 * gq + 0] * inv, o[4 * gq + 1] * inv);
                u.y = pack_bf2(o[4 * gq + 2] * inv, o[4 * gq + 3] * inv);
                *(uint2*)(od + 16 * gq) = u;
            }
        }
        if (g == 0) MV_SBA_STAMP();                       // 4: attention of group 0
        __syncthreads();
        if (g == 0) MV_SBA_STAMP();                       // 5: barrier
    }
    MV_SBA_STAMP();                                       // 6: groups 1, 2

    // ---------------- proj over the attention output ---------------------------------------------------------------------------------
    gemm(LDS_O, tile_base(p.wp, ta, NCT), tile_base(p.wp, ta + 1, NCT), tile_base(p.wp, ta, NCT), tile_base(p.wp, ta + 1, NCT));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int tile = u_tile[i], tb = u_tb[i];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int ch = 32 * tile + 8 * gq + 4 * fh;
            const float4 bv = *(const float4*)(p.bp + ch);
            *(float4*)(smem + (32 * tb + fr) * YROW + ch * 4) =
                make_float4(acc[i][4 * gq + 0] + bv.x, acc[i][4 * gq + 1] + bv.y, acc[i][4 * gq + 2] + bv.z, acc[i][4 * gq + 3] + bv.w);
        }
    }
    __syncthreads();
    MV_SBA_STAMP();                                       // 7: proj + staging + barrier
    // ---------------- epilogue: whole rows, + residual, back to the tokens' own positions -----------------------------------------------
    {
        constexpr int QPR = C / 4;
        constexpr int TOT = NWIN * NTOK * QPR;
        constexpr int NIT = (TOT + NT - 1) / NT;
        float4 xr[NIT];
        long long rows[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            int idx = tid + NT * i;
            idx = idx < TOT ? idx : TOT - 1;
            const int rr = idx / QPR, q = idx - rr * QPR;                 // rr: real token index over the windows
            const int r = 64 * (rr / NTOK) + rr % NTOK;
            rows[i] = tok_row(r) * C + 4 * q;
            xr[i] = *(const float4*)(p.x + rows[i]);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + NT * i;
            if (idx < TOT) {
                const int rr = idx / QPR, q = idx - rr * QPR;
                const int r = 64 * (rr / NTOK) + rr % NTOK;
                const float4 a = *(const float4*)(smem + r * YROW + q * 16);
                *(float4*)(p.y + rows[i]) = make_float4(a.x + xr[i].x, a.y + xr[i].y, a.z + xr[i].z, a.w + xr[i].w);
            }
        }
    }
    if (p.prof) {
        const long long tend = wall_clock64();
        if (lane == 0) {
            long long* o = p.prof + ((size_t)blockIdx.x * NWV + wave) * 9;
            for (int i = 0; i < 8; ++i) o[i] = st_w[i];
            o[8] = tend;
        }
    }
#undef MV_SBA_STAMP
}


// ---------------------------------------------------------------------------------------------------------------------------------
// C = 96 (stage 0: 3 heads, 64 windows per image): one workgroup = ONE window = two waves, and a wave OWNS a block of 32 tokens from
// the gather to the store -- the 8-wave kernel above spends 22 of its 24 us per workgroup in dependent latencies (gather -> LDS ->
// barrier, 8 barriers, a proj GEMM phase, an fp32 staging tile, a residual re-read) around 1.5 us of matrix work:
//   * a lane loads its token's row straight into the B-fragment layout (lane (fr, fh): channels 16 j + 8 fh .. + 7 of token fr, as
//     ln_mlp.hip); LayerNorm statistics are one lane^32 exchange; the normalised rows never touch LDS;
//   * the fp32 row it just read IS the residual: one v_permlane32_swap per register pair puts it in the accumulator layout, where it
//     seeds the proj accumulators -- no re-read in the epilogue (PMC of the 8-wave kernel: 115 MB read for 77 MB of rows);
//   * q | k | v of ALL three heads in one pass (9 tiles x 6 k-steps, weight fragments from L2 seven ahead); q stays in registers (its
//     accumulator layout is a valid B operand of S^T = K . Q^T once K is stored in the same channel order), k and v^T go to LDS:
//     29 KB per window, ONE barrier per workgroup;
//   * per head: S^T, bias + mask + softmax in registers, P . V, and the head's slice of proj straight from the P . V accumulators
//     (y^T += Wp[:, head] . o^T; the proj fragments arrive in the standard order and are re-ordered to the accumulator's channel
//     order by two half-wave swaps each) -- the attention output never exists in memory;
//   * the wave's 32 x 96 result leaves through LDS (K / V are dead by then) as whole 384-byte rows.
template <int UNUSED>
__global__ __launch_bounds__(128, 3) void swin_win96_kernel(const SwinBAP p) {
    constexpr int C = 96, KS = 6, NH = 3, NTOK = 49, WS = 7, QROW = 80, VROW = 144;
    constexpr int LDS_K = 0, LDS_V = NH * 64 * QROW, LDS_REG = LDS_V + NH * 32 * VROW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);              // = token block of the window
    const int fr = lane & 31, fh = lane >> 5;
    const int b = blockIdx.x / p.nW, wloc = blockIdx.x - b * p.nW;
#ifdef MV_W96_STAMPS          // per-wave phase stamps: 16 more registers, the kernel spills at three waves per SIMD -- build with -DMV_W96_STAMPS
    long long st_w[8];       // and __launch_bounds__(128, 2) to take them (profiles/r04/swin_c96_attn_pmc_new_vs_old.txt was taken that way)
    int nst = 0;
#define MV_W96_STAMP() do { if (p.prof && nst < 8) st_w[nst++] = wall_clock64(); } while (0)
#else
#define MV_W96_STAMP() do { } while (0)
#endif
    MV_W96_STAMP();                                       // 0: start
    const int wy = wloc / p.nWw, wx = wloc - wy * p.nWw;
    const bool shifted = (p.shh + p.shw) > 0;
    const int tok = 32 * wave + fr;
    const bool real = tok < NTOK;
    long long row;
    {
        const int t = real ? tok : 0;
        const int ty = t / WS, tx = t - ty * WS;
        int oy = wy * WS + ty + p.shh, ox = wx * WS + tx + p.shw;
        if (oy >= p.Hf) oy -= p.Hf;
        if (ox >= p.Wf) ox -= p.Wf;
        row = (((long long)b * p.Hf + oy) * p.Wf + ox) * C;
    }
    // ---- the first weight tile is on its way before anything else
    const uint4* wq = (const uint4*)p.wqkv + lane;                       // [head][q | k | v][k-step][lane]
    constexpr int NF = 3 * NH * KS, RING = 8, LOOK = 7;                 // 54 fragments of 1 KB, 7 in flight
    uint4 wb[RING];
#pragma unroll
    for (int i = 0; i < LOOK; ++i) wb[i] = wq[i * 64];

    // ---- gather + LayerNorm: B fragments of the qkv GEMM, and the residual in accumulator layout
    float raw[KS][8];
    {
#ifdef MV_I8_PROF
        const float* src = p.x + ((p.ablate & 8) ? (long long)(lane * 96) : row) + 8 * fh;
#else
        const float* src = p.x + row + 8 * fh;
#endif
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const float4 a = *(const float4*)(src + 16 * j), c = *(const float4*)(src + 16 * j + 4);
            raw[j][0] = a.x; raw[j][1] = a.y; raw[j][2] = a.z; raw[j][3] = a.w;
            raw[j][4] = c.x; raw[j][5] = c.y; raw[j][6] = c.z; raw[j][7] = c.w;
        }
    }
    if (tid < 64) {                                      // shift-mask region of every token (rolled coordinates, swin.py:190-209)
        const int t = tid < NTOK ? tid : NTOK - 1;
        const int ty = t / WS, tx = t - ty * WS;
        const int yy = wy * WS + ty, xx = wx * WS + tx;
        const int rh = (yy < p.Hf - WS) ? 0 : (yy < p.Hf - p.shh ? 1 : 2);
        const int rw = (xx < p.Wf - WS) ? 0 : (xx < p.Wf - p.shw ? 1 : 2);
        ((int*)(smem + LDS_REG))[tid] = shifted ? rh * 3 + rw : 0;
    }
    uint4 xf[KS];
    {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += raw[j][e];
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = raw[j][e] - mean;
                q = fmaf(d, d, q);
            }
        q += __shfl_xor(q, 32);
        const float rstd = real ? rsqrtf(q * (1.0f / C) + p.eps) : 0.f;                 // padded tokens: zero rows
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            xf[j].x = pack_bf2((raw[j][0] - mean) * rstd, (raw[j][1] - mean) * rstd);
            xf[j].y = pack_bf2((raw[j][2] - mean) * rstd, (raw[j][3] - mean) * rstd);
            xf[j].z = pack_bf2((raw[j][4] - mean) * rstd, (raw[j][5] - mean) * rstd);
            xf[j].w = pack_bf2((raw[j][6] - mean) * rstd, (raw[j][7] - mean) * rstd);
        }
    }
    // yacc[ct][4 g + i] = y^T[32 ct + 8 g + 4 fh + i][my token], seeded with the residual: the lane holds channels 16 j + 8 fh + 0..7;
    // after the swap the first register of a pair is group g' = 2 j, the second g' = 2 j + 1 (g' = 4 ct + g), on both halves
    f32x16 yacc[3];
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(raw[j][e]), __float_as_uint(raw[j][e + 4]), false, false);
            const int ga = 2 * j, gb = 2 * j + 1;
            yacc[ga >> 2][4 * (ga & 3) + e] = __uint_as_float(sw[0]);
            yacc[gb >> 2][4 * (gb & 3) + e] = __uint_as_float(sw[1]);
        }

    MV_W96_STAMP();                                       // 1: gather + LayerNorm + residual seed
    // ---- q | k | v of the three heads: tile u = 3 head + kind; a ring of 8 weight fragments, 7 loads in flight
    uint4 qf[NH][2];
#pragma unroll
    for (int u = 0; u < 3 * NH; ++u) {
        const int h = u / 3, kind = u - 3 * h;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int i = u * KS + j;
            const uint4 a = wb[i % RING];
#ifdef MV_I8_PROF
            if (i + LOOK < NF) wb[(i + LOOK) % RING] = wq[((p.ablate & 1) ? 0 : (i + LOOK)) * 64];
#else
            if (i + LOOK < NF) wb[(i + LOOK) % RING] = wq[(i + LOOK) * 64];
#endif
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, xf[j]), acc, 0, 0, 0);
        }
        const float* bb = p.bqkv + u * 32 + 4 * fh;
        float v[16];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 bv = *(const float4*)(bb + 8 * gq);
            v[4 * gq + 0] = acc[4 * gq + 0] + bv.x; v[4 * gq + 1] = acc[4 * gq + 1] + bv.y;
            v[4 * gq + 2] = acc[4 * gq + 2] + bv.z; v[4 * gq + 3] = acc[4 * gq + 3] + bv.w;
        }
        if (kind < 2) {
            // head channels in accumulator order: k-step j of a lane = channels 8 (2 j + (e >> 2)) + 4 fh + (e & 3) -- the same on the
            // q side (registers) and the k side (LDS row of the token, 16 bytes at (2 fh + j) * 16), so S^T = K . Q^T is exact
            uint4 pk[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                pk[j].x = pack_bf2(v[8 * j + 0], v[8 * j + 1]); pk[j].y = pack_bf2(v[8 * j + 2], v[8 * j + 3]);
                pk[j].z = pack_bf2(v[8 * j + 4], v[8 * j + 5]); pk[j].w = pack_bf2(v[8 * j + 6], v[8 * j + 7]);
            }
            if (kind == 0) { qf[h][0] = pk[0]; qf[h][1] = pk[1]; }
            else {
                char* dst = smem + LDS_K + (h * 64 + tok) * QROW + fh * 32;
                *(uint4*)dst = pk[0];
                *(uint4*)(dst + 16) = pk[1];
            }
        } else {
            // v^T[dh][slot(key)]: the 8 keys a lane feeds to one P.V step sit next to each other (same order as the 8-wave kernel)
            const int kg = fr >> 3, khh = (fr >> 2) & 1, kr = fr & 3;
            const int slot = ((wave * 2 + (kg >> 1)) * 2 + khh) * 8 + (kg & 1) * 4 + kr;
            char* dst = smem + LDS_V + h * 32 * VROW + slot * 2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
#pragma unroll
                for (int e = 0; e < 4; ++e) *(bf16_t*)(dst + (8 * gq + 4 * fh + e) * VROW) = f2bf(v[4 * gq + e]);
        }
    }
    MV_W96_STAMP();                                       // 2: qkv of three heads + k / v stores
    __syncthreads();
    MV_W96_STAMP();                                       // 3: barrier

    // ---- attention + the head's slice of proj
    const int* rl = (const int*)(smem + LDS_REG);
    const int qreg = rl[tok];
    const float scale = rsqrtf(32.0f);
    const uint4* wpq = (const uint4*)p.wp + lane;                        // [channel tile][k-step 0..5][lane]: k-steps 2 h, 2 h + 1 = head h
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        uint4 wpf[3][2];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
#ifdef MV_I8_PROF
            for (int j = 0; j < 2; ++j) wpf[ct][j] = wpq[((p.ablate & 1) ? 0 : (ct * KS + 2 * h + j)) * 64];
#else
            for (int j = 0; j < 2; ++j) wpf[ct][j] = wpq[(ct * KS + 2 * h + j) * 64];
#endif
        const char* kp = smem + LDS_K + (h * 64 + fr) * QROW + fh * 32;
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x8 kf = *(const bf16x8*)(kp + kt * 32 * QROW + j * 16);
                s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, __builtin_bit_cast(bf16x8, qf[h][j]), s[kt], 0, 0, 0);
            }
        }
        const float* brow = p.bias + ((size_t)h * 64 + tok) * 64;
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int key0 = 32 * kt + 8 * gq + 4 * fh;
#ifdef MV_I8_PROF
                const float4 bv = *(const float4*)(brow + ((p.ablate & 2) ? 4 * fh : key0));
#else
                const float4 bv = *(const float4*)(brow + key0);
#endif
                const int4 kr = *(const int4*)(rl + key0);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                const int rr[4] = {kr.x, kr.y, kr.z, kr.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = fmaf(s[kt][4 * gq + e], scale, bb[e]);
                    if (rr[e] != qreg) v += -100.0f;
                    s[kt][4 * gq + e] = v;
                    mx = fmaxf(mx, v);
                }
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = __expf(s[kt][e] - mx);
                s[kt][e] = pe;
                sum += pe;
            }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.f / sum;
        f32x16 o;
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = 0.f;
        const char* vp = smem + LDS_V + (h * 32 + fr) * VROW + fh * 16;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint4 pf;
                pf.x = pack_bf2(s[kt][8 * gp + 0], s[kt][8 * gp + 1]);
                pf.y = pack_bf2(s[kt][8 * gp + 2], s[kt][8 * gp + 3]);
                pf.z = pack_bf2(s[kt][8 * gp + 4], s[kt][8 * gp + 5]);
                pf.w = pack_bf2(s[kt][8 * gp + 6], s[kt][8 * gp + 7]);
                const bf16x8 vf = *(const bf16x8*)(vp + (kt * 2 + gp) * 32);
                o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, __builtin_bit_cast(bf16x8, pf), o, 0, 0, 0);
            }
        // o[4 gq + i] = out[my token][32 h + 8 gq + 4 fh + i]: k-step j of proj = registers 8 j .. 8 j + 7.  The standard proj fragment
        // of a lane holds input channels 16 jj + 8 fh + 0..7; the accumulator order wants {0..3, 8..11} below and {4..7, 12..15} above
        uint4 of[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            of[j].x = pack_bf2(o[8 * j + 0] * inv, o[8 * j + 1] * inv); of[j].y = pack_bf2(o[8 * j + 2] * inv, o[8 * j + 3] * inv);
            of[j].z = pack_bf2(o[8 * j + 4] * inv, o[8 * j + 5] * inv); of[j].w = pack_bf2(o[8 * j + 6] * inv, o[8 * j + 7] * inv);
        }
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                uint4 a = wpf[ct][j];
                const auto s0 = __builtin_amdgcn_permlane32_swap(a.x, a.z, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(a.y, a.w, false, false);
                a.x = s0[0]; a.z = s0[1]; a.y = s1[0]; a.w = s1[1];
                yacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, of[j]), yacc[ct], 0, 0, 0);
            }
        MV_W96_STAMP();                                   // 4, 5, 6: head h (attention + proj slice)
    }
    // ---- epilogue: + proj bias, to the token's own position (window reverse + roll back = where it was gathered from).  The
    // accumulator layout would store 32 contiguous bytes per row and instruction (partial lines: 20 % of the kernel's time in the
    // ablation, profiles/r04/swin_win96_ablation.txt); the wave's 32 x 96 fp32 tile goes through LDS instead -- K and V are dead by
    // now -- and leaves as whole 384-byte rows, 16 bytes per lane.
    __syncthreads();                                     // both waves are done with K / V
    {
        constexpr int YP = 100;                          // floats per staged row (96 + 4: conflict-free 16-byte pieces)
        float* yt = (float*)smem + wave * (32 * YP + 64);
        long long* ro = (long long*)(yt + 32 * YP);      // row offset of each of the wave's 32 tokens, -1 = padding token
        const float* bp = p.bp + 4 * fh;
#pragma unroll
        for (int ct = 0; ct < 3; ++ct)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4*)(bp + 32 * ct + 8 * g);
                *(float4*)(yt + fr * YP + 32 * ct + 8 * g + 4 * fh) = make_float4(yacc[ct][4 * g + 0] + bv.x, yacc[ct][4 * g + 1] + bv.y,
                                                                                    yacc[ct][4 * g + 2] + bv.z, yacc[ct][4 * g + 3] + bv.w);
            }
        if (fh == 0) ro[fr] = real ? row : -1;
        wave_lds_fence();
#ifdef MV_I8_PROF
        if ((p.ablate & 4) && yacc[0][0] != 12345.678f) return;
#endif
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int idx = lane + 64 * i, r = idx / 24, q = idx - r * 24;
            const long long off = ro[r];
            if (off >= 0) *(float4*)(p.y + off + 4 * q) = *(const float4*)(yt + r * YP + 4 * q);
        }
    }
#ifdef MV_W96_STAMPS
    if (p.prof) {
        __builtin_amdgcn_s_waitcnt(0);
        const long long tend = wall_clock64();
        if (lane == 0) {
            long long* o = p.prof + ((size_t)blockIdx.x * 2 + wave) * 9;
            for (int i = 0; i < 7; ++i) o[i] = st_w[i];
            o[7] = tend;
        }
    }
#endif
#undef MV_W96_STAMP
}

}  // namespace

}  // namespace mv

extern "C" {

// windows per workgroup.  (C = 96 also ran as 4 windows / 8 waves, one workgroup per CU: -4 % on swin_t, removed in round 4.)
static int swin_block_attn_nwin(int C) { return C == 384 ? 1 : (C == 192 ? 2 : (C == 96 ? 2 : 0)); }

int mv_swin_block_attn_supported(int Hf, int Wf, int C, int heads, int wsh, int wsw, int x_dtype) {
    if (mv::get_flag("no_swin_block_attn")) return 0;
    const int nwin = swin_block_attn_nwin(C);
    if (!nwin || x_dtype != MV_F32 || heads * 32 != C || wsh != 7 || wsw != 7 || Hf % 7 || Wf % 7 || Hf < 14 || Wf < 14) return 0;
    return ((Hf / 7) * (Wf / 7)) % nwin == 0;
}

int mv_swin_block_attn_fwd(const void* x, const void* wqkv_f, const float* bqkv, const void* wp_f, const float* bp, const float* bias64,
                           void* y, int B, int Hf, int Wf, int C, int heads, int wsh, int wsw, int shh, int shw, float eps, int x_dtype,
                           mv_stream_t stream_) {
    using namespace mv;
    hipStream_t stream = (hipStream_t)stream_;
    MV_CHECK_ARG(x && wqkv_f && bqkv && wp_f && bp && bias64 && y, "mv_swin_block_attn_fwd: null argument");
    MV_CHECK_ARG(x != y && B > 0, "mv_swin_block_attn_fwd: not in place; B = %d", B);
    if (!mv_swin_block_attn_supported(Hf, Wf, C, heads, wsh, wsw, x_dtype) || shh < 0 || shh >= wsh || shw < 0 || shw >= wsw) {
        set_error("mv_swin_block_attn_fwd: unsupported %dx%dx%d heads %d window %dx%d shift %d,%d (ask mv_swin_block_attn_supported first)",
                  Hf, Wf, C, heads, wsh, wsw, shh, shw);
        return MV_E_UNSUPPORTED;
    }
    SwinBAP p;
    p.x = (const float*)x; p.wqkv = (const bf16_t*)wqkv_f; p.bqkv = bqkv; p.wp = (const bf16_t*)wp_f; p.bp = bp; p.bias = bias64;
    p.y = (float*)y; p.Hf = Hf; p.Wf = Wf; p.shh = shh; p.shw = shw; p.nWw = Wf / 7; p.nW = (Hf / 7) * (Wf / 7); p.eps = eps;
    p.prof = nullptr; p.ablate = 0;
#ifdef MV_I8_PROF              // debug build only
    p.ablate = get_flag("w96_ablate");
    if (get_flag("sba_prof"))
        p.prof = (long long*)(((unsigned long long)(unsigned)get_flag("prof_hi") << 32) | (unsigned)get_flag("prof_lo"));
#endif
    const int nwin = swin_block_attn_nwin(C);
    const int nwv = C == 96 ? 4 : 8;
    const int tr = 64 * nwin, nslot = nwv / 2;
    const int xbytes = C == 96 ? tr * 192 + (tr / 4) * 16 : tr * (C * 2 + 16);
    const int smem = 2 * xbytes + 2 * nslot * 64 * 80 + nslot * 32 * 144 + nwin * 256;
    const dim3 grid((unsigned)(B * (p.nW / nwin)));
#define MV_SBA_GO(CC, NW, WV)                                                                                        \
    do {                                                                                                             \
        auto kern = swin_block_attn_kernel<CC, NW, WV>;                                                              \
        static LdsAttrSite attr;                                                                                     \
        MV_HIP(attr.ensure((const void*)kern, smem));                                                                \
        hipLaunchKernelGGL(kern, grid, dim3(WV * 64), smem, stream, p);                                              \
    } while (0)
    if (C == 384) { set_kernel_name("swin_block_attn_c384"); MV_SBA_GO(384, 1, 8); }
    else if (C == 192) { set_kernel_name("swin_block_attn_c192"); MV_SBA_GO(192, 2, 8); }
    else if (get_flag("swin_c96_shared")) { set_kernel_name("swin_block_attn_c96"); MV_SBA_GO(96, 2, 4); }
    else {
        set_kernel_name("swin_win96");
        constexpr int LDS96 = 3 * 64 * 80 + 3 * 32 * 144 + 256;
        hipLaunchKernelGGL(swin_win96_kernel<0>, dim3((unsigned)(B * p.nW)), dim3(128), LDS96, stream, p);
    }
#undef MV_SBA_GO
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

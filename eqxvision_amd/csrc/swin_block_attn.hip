// The attention half of a Swin block in ONE launch, one workgroup per window (gfx950):
//     y = x + proj(window_attention(qkv(LayerNorm(x))))          (swin.py:572-578 first line, 90-255, 342-366)
// for C = 384 (stage 2 of swin_t / swin_s: 12 heads of 32, 7 x 7 windows, 14 x 14 maps -> 4 windows per image, one round of
// 256 workgroups at 64 images).  The un-fused path is four launches (LayerNorm, qkv Linear, window attention, proj Linear +
// residual) that move the 1152-channel qkv tensor and the attention output through HBM; here a window's 49 token rows are
// gathered once (cyclic shift + window partition = index arithmetic), and q / k / v / the attention output live in LDS:
//
//   phase 0   LayerNorm of the 49 rows (fp32 statistics) -> bf16 rows in LDS (rows 49..63 zero; LayerNorm affine folded into
//             the qkv weights by the host)
//   3 x       heads in groups of four: [q | k | v] of the group = 12 channel tiles x 2 token blocks = 24 MFMA tiles, three per
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
    const bf16_t* wqkv;    // [C/128 groups][12 tiles: q0..3 k0..3 v0..3][C/16][64][8]
    const float* bqkv;     // [groups][12][32]
    const bf16_t* wp;      // [C/32 tiles][C/16][64][8]
    const float* bp;       // [C]
    const float* bias;     // [heads][64][64]: relative-position bias, -1e30 on padded keys
    float* y;
    int Hf, Wf, shh, shw, nWw, nW;
    float eps;
};

template <int C>
__global__ __launch_bounds__(512) void swin_block_attn_kernel(const SwinBAP p) {
    constexpr int NH = C / 32, NG = NH / 4, KS = C / 16, NTOK = 49, WS = 7, D = 4;
    static_assert(NH % 4 == 0 && C % 48 == 0 && KS % D == 0, "C = 384 layout");
    constexpr int XROW = C * 2 + 16;                     // token rows of the normalised input / the attention output
    constexpr int QROW = 80, VROW = 144;                 // q, k rows: 32 dh (+ pad); v^T rows: 64 keys (+ pad)
    constexpr int LDS_NX = 0;
    constexpr int LDS_Q = 64 * XROW;                     // [4 heads][64 tokens][QROW]
    constexpr int LDS_K = LDS_Q + 4 * 64 * QROW;
    constexpr int LDS_V = LDS_K + 4 * 64 * QROW;         // [4 heads][32 dh][VROW]
    constexpr int LDS_O = LDS_V + 4 * 32 * VROW;
    constexpr int LDS_REG = LDS_O + 64 * XROW;           // int[64]: shift-mask region per token
    constexpr int YROW = C * 4 + 16;
    static_assert(64 * YROW <= LDS_O, "result tile must fit below the attention output");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int b = blockIdx.x / p.nW, wloc = blockIdx.x - b * p.nW;
    const int wy = wloc / p.nWw, wx = wloc - wy * p.nWw;
    const bool shifted = (p.shh + p.shw) > 0;
    auto tok_row = [&](int t) -> long long {            // token t of this window -> its row in x / y (un-rolled position)
        const int ty = t / WS, tx = t - ty * WS;
        int oy = wy * WS + ty + p.shh, ox = wx * WS + tx + p.shw;
        if (oy >= p.Hf) oy -= p.Hf;
        if (ox >= p.Wf) ox -= p.Wf;
        return ((long long)b * p.Hf + oy) * p.Wf + ox;
    };

    // ---------------- weight-fragment streams: two channel tiles per wave (units 3w .. 3w+2 of 12 tiles x 2 token blocks) ------
    const int ta = (3 * wave) >> 1;                                    // first tile; the second is ta + 1
    const bool odd = wave & 1;                                         // even: (ta,0) (ta,1) (ta+1,0); odd: (ta,1) (ta+1,0) (ta+1,1)
    auto tile_base = [&](const bf16_t* w, int tile) -> const uint4* { return (const uint4*)w + (size_t)tile * KS * 64 + lane; };
    uint4 a0[D], a1[D];
    {
        const uint4* s0 = tile_base(p.wqkv, ta);
        const uint4* s1 = tile_base(p.wqkv, ta + 1);
#pragma unroll
        for (int d = 0; d < D; ++d) { a0[d] = s0[d * 64]; a1[d] = s1[d * 64]; }
    }

    // ---------------- phase 0: gather + LayerNorm -> LDS ----------------------------------------------------------------------
    {
        constexpr int LPR = C / 12, RPP = 64 / LPR, NP = 8 / RPP;
        const int lr = lane / LPR, lq = lane % LPR;
#pragma unroll
        for (int ps = 0; ps < NP; ++ps) {
            const int r = 8 * wave + ps * RPP + lr;
            const float4* src = (const float4*)(p.x + tok_row(r < NTOK ? r : NTOK - 1) * C);
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
            const float rstd = r < NTOK ? rsqrtf(q * (1.0f / C) + p.eps) : 0.f;        // padded rows: zeros
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                uint2 u;
                u.x = pack_bf2(v[i].x * rstd, v[i].y * rstd);
                u.y = pack_bf2(v[i].z * rstd, v[i].w * rstd);
                *(uint2*)(smem + LDS_NX + r * XROW + (lq + LPR * i) * 8) = u;
            }
        }
        if (tid < 64) {                                  // shift-mask region of every token (rolled coordinates, swin.py:190-209)
            const int t = tid < NTOK ? tid : NTOK - 1;
            const int ty = t / WS, tx = t - ty * WS;
            const int yy = wy * WS + ty, xx = wx * WS + tx;
            const int rh = (yy < p.Hf - WS) ? 0 : (yy < p.Hf - p.shh ? 1 : 2);
            const int rw = (xx < p.Wf - WS) ? 0 : (xx < p.Wf - p.shw ? 1 : 2);
            ((int*)(smem + LDS_REG))[tid] = shifted ? rh * 3 + rw : 0;
        }
    }
    __syncthreads();

    // ---------------- the three-tiles-per-wave GEMM: acc[i] = unit i of this wave over K = C --------------------------------------
    f32x16 acc[3];
    // src: LDS byte offset of the token rows (B operand); cur0 / cur1: this phase's two fragment streams; nx0 / nx1: the next phase's
    auto gemm3 = [&](int src, const uint4* cur0, const uint4* cur1, const uint4* nx0, const uint4* nx1) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        const char* xb = smem + src + fr * XROW + fh * 16;
        bf16x8 bq[4][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) bq[t][tb] = *(const bf16x8*)(xb + tb * 32 * XROW + t * 32);
        for (int j0 = 0; j0 < KS; j0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int j = j0 + d;
                const int jn = j + 2 < KS ? j + 2 : KS - 1;
                const bool in = j + D < KS;
                const uint4* n0 = in ? cur0 + (size_t)(j + D) * 64 : nx0 + (size_t)(j + D - KS) * 64;
                const uint4* n1 = in ? cur1 + (size_t)(j + D) * 64 : nx1 + (size_t)(j + D - KS) * 64;
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 f0 = __builtin_bit_cast(bf16x8, a0[d]);
                const bf16x8 f1 = __builtin_bit_cast(bf16x8, a1[d]);
                a0[d] = *n0;
                a1[d] = *n1;
                const bf16x8 b0 = bq[d][0], b1 = bq[d][1];
                if (!odd) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, b1, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, b0, acc[2], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, b1, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, b0, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, b1, acc[2], 0, 0, 0);
                }
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) bq[(d + 2) % 4][tb] = *(const bf16x8*)(xb + tb * 32 * XROW + jn * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // unit i of this wave -> (tile, token block)
    auto unit_tile = [&](int i) -> int { return odd ? (i == 0 ? ta : ta + 1) : (i == 2 ? ta + 1 : ta); };
    auto unit_tb = [&](int i) -> int { return odd ? (i == 1 ? 0 : 1) : (i == 1 ? 1 : 0); };

    const int* regl = (const int*)(smem + LDS_REG);
    const float scale = rsqrtf(32.0f);

    for (int g = 0; g < NG; ++g) {
        // ---------------- q | k | v of heads 4g .. 4g+3 ----------------------------------------------------------------------------
        const bf16_t* wg = p.wqkv + (size_t)g * 12 * KS * 64 * 8;
        const bf16_t* wn = g + 1 < NG ? p.wqkv + (size_t)(g + 1) * 12 * KS * 64 * 8 : p.wp;      // next phase: next group, then proj
        gemm3(LDS_NX, tile_base(wg, ta), tile_base(wg, ta + 1), tile_base(wn, ta), tile_base(wn, ta + 1));
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int tile = unit_tile(i), tb = unit_tb(i);
            const int kind = tile >> 2, hl = tile & 3;                  // 0 q, 1 k, 2 v; head of the group
            const float* bb = p.bqkv + ((size_t)g * 12 + tile) * 32 + 4 * fh;
            const int tok = 32 * tb + fr;
            if (kind < 2) {
                char* dst = smem + (kind == 0 ? LDS_Q : LDS_K) + (hl * 64 + tok) * QROW + 8 * fh;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const float4 bv = *(const float4*)(bb + 8 * gq);
                    uint2 u;
                    u.x = pack_bf2(acc[i][4 * gq + 0] + bv.x, acc[i][4 * gq + 1] + bv.y);
                    u.y = pack_bf2(acc[i][4 * gq + 2] + bv.z, acc[i][4 * gq + 3] + bv.w);
                    *(uint2*)(dst + 16 * gq) = u;
                }
            } else {
                // v^T[dh][slot(key)]: the 8 keys a lane feeds to one P.V step sit next to each other (slot order (tile, pair of
                // groups, lane half, group parity, r)) -- the lane's own position decides the slot of its token
                const int kg = fr >> 3, khh = (fr >> 2) & 1, kr = fr & 3;
                const int slot = ((tb * 2 + (kg >> 1)) * 2 + khh) * 8 + (kg & 1) * 4 + kr;
                char* dst = smem + LDS_V + hl * 32 * VROW + slot * 2;
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
        // ---------------- attention: wave = (head hl of the group, query block qb) ------------------------------------------------
        {
            const int hl = wave >> 1, qb = wave & 1, h = 4 * g + hl;
            const int qtok = 32 * qb + fr;
            const char* qp = smem + LDS_Q + (hl * 64 + qtok) * QROW + fh * 16;
            const char* kp = smem + LDS_K + (hl * 64 + fr) * QROW + fh * 16;
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
            const int qreg = regl[qtok];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int key0 = 32 * kt + 8 * gq + 4 * fh;
                    const float4 bv = *(const float4*)(brow + key0);
                    const int4 kr = *(const int4*)(regl + key0);
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
            const char* vp = smem + LDS_V + (hl * 32 + fr) * VROW + fh * 16;
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
            char* od = smem + LDS_O + qtok * XROW + (32 * h + 4 * fh) * 2;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 u;
                u.x = pack_bf2(o[4 * gq + 0] * inv, o[4 * gq + 1] * inv);
                u.y = pack_bf2(o[4 * gq + 2] * inv, o[4 * gq + 3] * inv);
                *(uint2*)(od + 16 * gq) = u;
            }
        }
        __syncthreads();
    }

    // ---------------- proj over the attention output ---------------------------------------------------------------------------------
    gemm3(LDS_O, tile_base(p.wp, ta), tile_base(p.wp, ta + 1), tile_base(p.wp, ta), tile_base(p.wp, ta + 1));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int tile = unit_tile(i), tb = unit_tb(i);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int ch = 32 * tile + 8 * gq + 4 * fh;
            const float4 bv = *(const float4*)(p.bp + ch);
            *(float4*)(smem + (32 * tb + fr) * YROW + ch * 4) =
                make_float4(acc[i][4 * gq + 0] + bv.x, acc[i][4 * gq + 1] + bv.y, acc[i][4 * gq + 2] + bv.z, acc[i][4 * gq + 3] + bv.w);
        }
    }
    __syncthreads();
    // ---------------- epilogue: whole rows, + residual, back to the tokens' own positions -----------------------------------------------
    {
        constexpr int QPR = C / 4;
        constexpr int TOT = NTOK * QPR;
        constexpr int NIT = (TOT + 511) / 512;
        float4 xr[NIT];
        long long rows[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            int idx = tid + 512 * i;
            idx = idx < TOT ? idx : TOT - 1;
            const int r = idx / QPR, q = idx - r * QPR;
            rows[i] = tok_row(r) * C + 4 * q;
            xr[i] = *(const float4*)(p.x + rows[i]);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + 512 * i;
            if (idx < TOT) {
                const int r = idx / QPR, q = idx - r * QPR;
                const float4 a = *(const float4*)(smem + r * YROW + q * 16);
                *(float4*)(p.y + rows[i]) = make_float4(a.x + xr[i].x, a.y + xr[i].y, a.z + xr[i].z, a.w + xr[i].w);
            }
        }
    }
}

}  // namespace

}  // namespace mv

extern "C" {

int mv_swin_block_attn_supported(int Hf, int Wf, int C, int heads, int wsh, int wsw, int x_dtype) {
    if (mv::get_flag("no_swin_block_attn")) return 0;
    return x_dtype == MV_F32 && C == 384 && heads == 12 && wsh == 7 && wsw == 7 && Hf % 7 == 0 && Wf % 7 == 0 && Hf >= 14 && Wf >= 14;
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
    constexpr int SMEM = 2 * 64 * (384 * 2 + 16) + 2 * 4 * 64 * 80 + 4 * 32 * 144 + 256;
    auto kern = swin_block_attn_kernel<384>;
    MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    set_kernel_name("swin_block_attn_c384");
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * p.nW)), dim3(512), SMEM, stream, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

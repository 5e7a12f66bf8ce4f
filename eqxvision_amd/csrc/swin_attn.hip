// Shifted-window attention core for gfx950 (reference `_shifted_window_attention`, swin.py:123-250) on the
// qkv Linear2d output: qkv NHWC [B,Hf,Wf,3C] (channel order [q|k|v][head][dh]) -> out NHWC [B,Hf,Wf,C].
//
// MFMA path for the Swin-T/S/B shapes: windows of <= 64 tokens (7x7 = 49), dh = 32.
//   * cyclic shift, window partition / reverse are index arithmetic: the token at rolled position p lives at
//     original position (p + shift) mod size, and its output goes back to the same original position;
//   * one wave per (window, head): K and Q fragments (16 bytes = 8 channels of one token) come straight
//     from global memory -- nothing is shared between waves except the head's bias table;
//   * S^T = K.Q^T puts one query per lane: bias (+ shift mask, -100 like the reference) is added and the
//     softmax is done on accumulator registers + one lane^32 exchange;
//   * V is transposed through a wave-private LDS patch (key-contiguous) so the PV MFMA reads 8-byte fragments;
//     the PV reduction index uses the accumulator's own register order (no cross-lane movement of P);
//   * block = 4 waves sharing one head: the head's relative-position bias, padded to 64x64 fp32, is staged in
//     LDS once per block and re-used for all the windows the block walks.
#include "mfma_common.h"
#include "rng_common.h"

namespace mv {

struct SwinP {
    const bf16_t* qkv;
    const float* bias;     // [heads][n][n]
    bf16_t* out;
    int B, Hf, Wf, C, heads, wsh, wsw, shh, shw;
    int n, nWw, nW, total_windows;
    const uint32_t* drop_keys;   // DROP: one Threefry key per sample, [B][2]
    float keep;
};

// DROP: the reference's `_func_dropout(attn, attention_dropout, key)` (swin.py:227, applied in EVERY mode): probability
// (window w, head h, query i, key j) of a sample is kept (/ keep) or zeroed by word ((w * heads + h) * n + i) * n + j of the
// sample's Threefry stream = jax.random.bernoulli(key, keep, (nW, heads, n, n)), between the softmax and P . V.
template <bool DROP>
__global__ __launch_bounds__(256) void swin_attn_mfma_kernel(const SwinP p) {
    constexpr int DH = 32;
    constexpr int VPITCH = 64 * 2 + 8;                  // bytes per V^T row (64 keys + pad): 8 * odd
    __shared__ __attribute__((aligned(16))) float bias_l[64 * 64];        // [q][key], padded, this block's head
    __shared__ __attribute__((aligned(16))) char vt_l[4][DH * VPITCH];    // per wave V^T
    __shared__ __attribute__((aligned(16))) int kreg_l[4][64];                                         // per wave: shift-mask region id per key

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.y;
    const int n = p.n;
    {   // 16 entries per thread, all loads in flight before the first LDS store (one dependent round trip, not 16)
        float bv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int i = j * 256 + tid, q = i >> 6, k = i & 63;
            const bool ok = q < n && k < n;
            const float t = p.bias[ok ? ((long long)h * n + q) * n + k : 0];
            bv[j] = ok ? t : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) bias_l[j * 256 + tid] = bv[j];
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const long long rs = 3LL * p.C;
    const bool shifted = (p.shh + p.shw) > 0;
    const float scale = rsqrtf((float)DH);
    char* vt = vt_l[wave];
    int* kreg = kreg_l[wave];

    for (int win = blockIdx.x * 4 + wave; win < p.total_windows; win += gridDim.x * 4) {
        const int b = win / p.nW, wloc = win - b * p.nW;
        const int wy = wloc / p.nWw, wx = wloc - wy * p.nWw;
        // token t of this window -> element offset of its qkv row (original, un-rolled position)
        auto tok_row = [&](int t) -> long long {
            const int ty = t / p.wsw, tx = t - ty * p.wsw;
            int oy = wy * p.wsh + ty + p.shh, ox = wx * p.wsw + tx + p.shw;
            if (oy >= p.Hf) oy -= p.Hf;
            if (ox >= p.Wf) ox -= p.Wf;
            return ((long long)b * p.Hf + oy) * p.Wf + ox;
        };
        auto region = [&](int t) {
            const int ty = t / p.wsw, tx = t - ty * p.wsw;
            const int y = wy * p.wsh + ty, x = wx * p.wsw + tx;          // rolled coordinates
            const int rh = (y < p.Hf - p.wsh) ? 0 : (y < p.Hf - p.shh ? 1 : 2);
            const int rw = (x < p.Wf - p.wsw) ? 0 : (x < p.Wf - p.shw ? 1 : 2);
            return rh * 3 + rw;
        };
        if (shifted) kreg[lane] = lane < n ? region(lane) : -1;

        // ---- V^T into LDS: each lane transposes 2 keys x 8 d (two items per lane cover 64 keys x 32 d)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = it * 64 + lane;              // 0..127 : (chunk ch = item / 32, key pair kp = item % 32)
            const int ch = item >> 5, key = 2 * (item & 31);
            uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
            const int k0 = key < n ? key : n - 1, k1 = key + 1 < n ? key + 1 : n - 1;
            const uint4 l0 = *(const uint4*)(p.qkv + tok_row(k0) * rs + 2 * p.C + h * DH + ch * 8);
            const uint4 l1 = *(const uint4*)(p.qkv + tok_row(k1) * rs + 2 * p.C + h * DH + ch * 8);
            if (key < n) v0 = l0;
            if (key + 1 < n) v1 = l1;
            const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w}, c[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                *(uint32_t*)(vt + (ch * 8 + 2 * e) * VPITCH + key * 2) = (a[e] & 0xffffu) | (c[e] << 16);
                *(uint32_t*)(vt + (ch * 8 + 2 * e + 1) * VPITCH + key * 2) = (a[e] >> 16) | (c[e] & 0xffff0000u);
            }
        }

        wave_lds_fence();      // kreg / V^T written by the wave's lanes above, read by other lanes below

        // ---- K and Q fragments straight from global: token = 32*tile + fr, channels 16*kk + 8*fh .. +7
        uint4 kf[2][2], qf[2][2];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const int t = tl * 32 + fr;
            const int tc = t < n ? t : n - 1;
            const bf16_t* row = p.qkv + tok_row(tc) * rs + h * DH + fh * 8;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                qf[tl][kk] = *(const uint4*)(row + kk * 16);
                kf[tl][kk] = *(const uint4*)(row + p.C + kk * 16);
            }
        }
        // my query's region id and output row (q-tile tq: query = 32*tq + fr)
        int qreg[2];
        long long orow[2];
#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
            const int q = tq * 32 + fr;
            const int qc = q < n ? q : n - 1;
            qreg[tq] = shifted ? region(qc) : 0;
            orow[tq] = tok_row(qc);
        }

#pragma unroll
        for (int tq = 0; tq < 2; ++tq) {
            const int q = tq * 32 + fr;
            // S^T[key][q]: two key tiles
            f32x16 s[2];
#pragma unroll
            for (int tk = 0; tk < 2; ++tk) {
#pragma unroll
                for (int e = 0; e < 16; ++e) s[tk][e] = 0.f;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    s[tk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[tk][kk]),
                                                                    __builtin_bit_cast(bf16x8, qf[tq][kk]), s[tk], 0, 0, 0);
            }
            // scale, bias, mask; accumulator register e of key tile tk = key 32*tk + (e&3) + 8*(e>>2) + 4*fh
            float mx = -INFINITY;
#pragma unroll
            for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int key0 = tk * 32 + 8 * g + 4 * fh;
                    const float4 bv = *(const float4*)(bias_l + (q & 63) * 64 + key0);
                    int4 kr = make_int4(0, 0, 0, 0);
                    if (shifted) kr = *(const int4*)(kreg + key0);
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
                    const int rr[4] = {kr.x, kr.y, kr.z, kr.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = s[tk][4 * g + e] * scale + bb[e];
                        if (shifted && rr[e] != qreg[tq]) v += -100.0f;
                        if (key0 + e >= n) v = -INFINITY;
                        s[tk][4 * g + e] = v;
                        mx = fmaxf(mx, v);
                    }
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float pe = __expf(s[tk][e] - mx);
                    s[tk][e] = pe;
                    sum += pe;
                }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.f / sum;
            if constexpr (DROP) {
                const uint32_t k0 = p.drop_keys[2 * b], k1 = p.drop_keys[2 * b + 1];
                const uint32_t nel = (uint32_t)p.nW * p.heads * n * n;
                const uint32_t base = (((uint32_t)wloc * p.heads + h) * n + (q < n ? q : 0)) * n;
                const float rk = 1.f / p.keep;
#pragma unroll
                for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = tk * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                        const bool on = word_uniform01(stream_word(k0, k1, base + (key < n ? key : 0), nel)) < p.keep;
                        s[tk][e] = on ? s[tk][e] * rk : 0.f;
                    }
            }
            // O^T[d][q] = sum_key V^T[d][key] P[key][q]
            f32x16 o;
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = 0.f;
#pragma unroll
            for (int tk = 0; tk < 2; ++tk)
#pragma unroll
                for (int t2 = 0; t2 < 2; ++t2) {
                    uint4 pf;
                    pf.x = pack_bf2(s[tk][8 * t2 + 0], s[tk][8 * t2 + 1]);
                    pf.y = pack_bf2(s[tk][8 * t2 + 2], s[tk][8 * t2 + 3]);
                    pf.z = pack_bf2(s[tk][8 * t2 + 4], s[tk][8 * t2 + 5]);
                    pf.w = pack_bf2(s[tk][8 * t2 + 6], s[tk][8 * t2 + 7]);
                    const int key0 = tk * 32 + 16 * t2 + 4 * fh;
                    const char* vrow = vt + fr * VPITCH + key0 * 2;
                    const uint2 lo = *(const uint2*)(vrow);
                    const uint2 hi = *(const uint2*)(vrow + 16);
                    const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf),
                                                                o, 0, 0, 0);
                }
            if (q < n) {
                bf16_t* dst = p.out + orow[tq] * p.C + h * DH;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d = 8 * g + 4 * fh;
                    uint2 u;
                    u.x = pack_bf2(o[4 * g] * inv, o[4 * g + 1] * inv);
                    u.y = pack_bf2(o[4 * g + 2] * inv, o[4 * g + 3] * inv);
                    *(uint2*)(dst + d) = u;
                }
            }
        }
        wave_lds_fence();      // the next window's kreg / V^T overwrite what other lanes may still be reading
    }
}

int swin_mfma_supported(int C, int heads, int ws_h, int ws_w, int dtype) {
    return dtype == MV_BF16 && heads > 0 && C / heads == 32 && C % heads == 0 && ws_h * ws_w <= 64 && ws_h * ws_w >= 2;
}

int swin_mfma_launch(const void* qkv, const float* bias, void* out, int B, int Hf, int Wf, int C, int heads, int ws_h,
                     int ws_w, int shift_h, int shift_w, const uint32_t* drop_keys, float keep, hipStream_t st) {
    SwinP p;
    p.drop_keys = drop_keys; p.keep = keep;
    p.qkv = (const bf16_t*)qkv; p.bias = bias; p.out = (bf16_t*)out;
    p.B = B; p.Hf = Hf; p.Wf = Wf; p.C = C; p.heads = heads; p.wsh = ws_h; p.wsw = ws_w; p.shh = shift_h; p.shw = shift_w;
    p.n = ws_h * ws_w;
    p.nWw = Wf / ws_w;
    p.nW = (Hf / ws_h) * p.nWw;
    const long long tw = (long long)B * p.nW;
    if (tw >= (1LL << 31)) {
        set_error("swin_attn: too many windows");
        return MV_E_UNSUPPORTED;
    }
    p.total_windows = (int)tw;
    int gx = (int)((tw + 3) / 4);
    // persistent blocks: two per CU in total (register-limited residency), so the per-block bias staging (16 KB) is
    // amortised over many windows (with 8 blocks per CU a wave saw ~1.5 windows: 66 -> 41 us on the 56x56 stage)
    const int per_cu = 2;
    const int cap = (256 * per_cu) / heads > 0 ? (256 * per_cu) / heads : 1;
    if (gx > cap) gx = cap;
    if (drop_keys && (long long)p.nW * heads * p.n * p.n >= (1LL << 32)) {
        set_error("swin_attn: more than 2^32 probabilities per sample");
        return MV_E_UNSUPPORTED;
    }
    set_kernel_name(drop_keys ? "swin_attn_mfma_dropout" : "swin_attn_mfma");
    if (drop_keys) hipLaunchKernelGGL(swin_attn_mfma_kernel<true>, dim3(gx, heads), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(swin_attn_mfma_kernel<false>, dim3(gx, heads), dim3(256), 0, st, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // namespace mv

// Fused LayerNorm -> fc1 -> GELU(tanh) -> fc2 -> + residual for narrow rows (C = 96, hidden = 384: the MLP half of the
// Swin stage-0 block, reference swin.py:572-578 `x + mlp(norm2(x))`, mlps.py:54-66), gfx950.
//
//   y[m, :] = x[m, :] + W2 . gelu( W1 . n(x[m, :]) + b1 ) + b2,      n(x) = (x - mean(x)) * rsqrt(var(x) + eps)
//
// (the LayerNorm affine is folded into W1 / b1 by the caller).  At M = 200 704 rows per 64 images the three separate launches
// move 115 + 193 + 308 MB (fp32 residual stream): the hidden activations alone are 154 MB written and read back.  Here both
// weight matrices (2 x 72 KB bf16) stay in LDS for the whole launch and the hidden activations never leave registers:
//   * every wave owns whole 32-row tiles; a lane loads its share of the rows straight from HBM (fp32 or bf16), the row statistics
//     are two cross-half shuffles, and the normalised values become the B fragments of fc1 directly;
//   * fc1 is computed TRANSPOSED (h^T = W1 . n(x)^T), 32 hidden units at a time: in the accumulator layout a lane then holds
//     8 + 8 hidden values of ITS row -- which is exactly the B-fragment shape of the next MFMA, so after bias + GELU + bf16
//     packing they feed fc2 (y^T += W2[:, chunk] . h_chunk^T) without any LDS transpose.  W2 is stored in LDS with its hidden
//     axis permuted to the order the accumulator registers come in;
//   * no block barrier after the weight fill: waves free-run over a strided tile list, loads / MFMA / GELU (the VALU-bound part:
//     77 M activations per launch) of different waves overlap;
//   * traffic: x once (+ an L2-hot re-read for the residual add), y once: 154 MB instead of 616 MB.
#include "mfma_common.h"

namespace mv {

struct LnMlpP {
    const void* x;
    const bf16_t* w1;     // [384][96], LayerNorm gamma folded in
    const float* b1;      // [384], b1 + W1 . beta
    const bf16_t* w2;     // [96][384]
    const float* b2;      // [96]
    void* y;
    int M, tiles;
    float eps;
    int dbg;              // MV_I8_PROF builds only: 1 = skip GELU, 2 = skip fc2 MFMAs (ablation timing)
};

namespace lm {
constexpr int C = 96, H = 384, KC = C / 16, CH = H / 32, RB = C / 32;
constexpr int W1P = C * 2 + 16;            // 208 B: odd number of 16-byte slots -> conflict-free fragment reads
constexpr int W2P = H * 2 + 16;            // 784 B
constexpr int W1B = H * W1P, W2B = C * W2P;
constexpr int SMEM = W1B + W2B + H * 4 + C * 4;
}  // namespace lm

template <typename XT> struct LmRow;
template <> struct LmRow<float> {
    __device__ static __forceinline__ void ld8(const float* p, float* v) {
        const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
};
template <> struct LmRow<bf16_t> {
    __device__ static __forceinline__ void ld8(const bf16_t* p, float* v) {
        const uint4 u = *(const uint4*)p;
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
        v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
    }
};

template <typename XT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void ln_mlp96_kernel(const LnMlpP p) {
    using namespace lm;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* w1l = smem;
    char* w2l = smem + W1B;
    float* b1l = (float*)(w2l + W2B);
    float* b2l = b1l + H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NT = WAVES * 64;
    static_assert(NT >= H, "bias fill assumes one thread per hidden unit");

    // ---- weights -> LDS, once.  W1 row-major; W2 with the hidden axis in accumulator order: 16-byte slot
    // q = (chunk*2 + s)*2 + half of row c holds hidden {32*chunk + 16*s + 4*half + 0..3, the same + 8}.
    // All global loads of a thread are issued before its first LDS store (one L2 round trip, not one per piece).
    {
        constexpr int N1 = H * (C / 8), K1 = (N1 + NT - 1) / NT;
        constexpr int N2 = C * (H / 4), K2 = (N2 + NT - 1) / NT;
        uint4 t1[K1];
        uint2 t2[K2];
#pragma unroll
        for (int k = 0; k < K1; ++k) {
            const int i = tid + k * NT, r = i / (C / 8), c = i - r * (C / 8);
            if (i < N1) t1[k] = *(const uint4*)(p.w1 + r * C + c * 8);
        }
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            const int i = tid + k * NT, r = i / (H / 4), d = i - r * (H / 4);
            const int slot = d >> 1, e = d & 1, half = slot & 1, js = slot >> 1;
            if (i < N2) t2[k] = *(const uint2*)(p.w2 + r * H + 16 * js + 8 * e + 4 * half);
        }
        const float bb1 = tid < H ? p.b1[tid] : 0.f, bb2 = tid < C ? p.b2[tid] : 0.f;
#pragma unroll
        for (int k = 0; k < K1; ++k) {
            const int i = tid + k * NT, r = i / (C / 8), c = i - r * (C / 8);
            if (i < N1) *(uint4*)(w1l + r * W1P + c * 16) = t1[k];
        }
#pragma unroll
        for (int k = 0; k < K2; ++k) {
            const int i = tid + k * NT, r = i / (H / 4), d = i - r * (H / 4);
            if (i < N2) *(uint2*)(w2l + r * W2P + d * 8) = t2[k];
        }
        if (tid < H) b1l[tid] = bb1;
        if (tid < C) b2l[tid] = bb2;
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const XT* x = (const XT*)p.x;
    XT* y = (XT*)p.y;
    const char* w1f = w1l + fr * W1P + fh * 16;        // + chunk*32*W1P + t*32
    const char* w2f = w2l + fr * W2P + fh * 16;        // + r*32*W2P + (chunk*2 + s)*32
    const float* b1f = b1l + 4 * fh;                   // + 32*chunk + 8*g
    const float* b2f = b2l + 4 * fh;                   // + 32*r + 8*g

    float raw[KC][8];
    auto load_raw = [&](int tile) {
        int m = tile * 32 + fr;
        m = m < p.M ? m : p.M - 1;                     // clamp: rows past the end are never stored
        const XT* src = x + (long long)m * C + fh * 8;
#pragma unroll
        for (int t = 0; t < KC; ++t) LmRow<XT>::ld8(src + t * 16, raw[t]);
    };

    const int stride = gridDim.x * WAVES;
    int tile = wave * gridDim.x + blockIdx.x;          // consecutive tiles go to different CUs
    constexpr bool PREFETCH = WAVES <= 8;              // 12+ waves per CU: no room for a second row set
    if (PREFETCH && tile < p.tiles) load_raw(tile);
    for (; tile < p.tiles; tile += stride) {
        if (!PREFETCH) load_raw(tile);
        // ---- LayerNorm statistics of my row: 48 values here, 48 in lane ^ 32
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < KC; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += raw[t][e];
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < KC; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = raw[t][e] - mean;
                q = fmaf(d, d, q);
            }
        q += __shfl_xor(q, 32);
        const float rstd = rsqrtf(q * (1.0f / C) + p.eps);
        uint4 xf[KC];
#pragma unroll
        for (int t = 0; t < KC; ++t) {
            xf[t].x = pack_bf2((raw[t][0] - mean) * rstd, (raw[t][1] - mean) * rstd);
            xf[t].y = pack_bf2((raw[t][2] - mean) * rstd, (raw[t][3] - mean) * rstd);
            xf[t].z = pack_bf2((raw[t][4] - mean) * rstd, (raw[t][5] - mean) * rstd);
            xf[t].w = pack_bf2((raw[t][6] - mean) * rstd, (raw[t][7] - mean) * rstd);
        }
        // ---- the fc2 accumulators START as the residual: a lane needs channels 8 g' + 4 fh + 0..3 (g' = 4 r + g) of its row and
        // holds 16 t + 8 fh + 0..7; one v_permlane32_swap per register pair -- lanes 32..63 of the low half-group trade places with
        // lanes 0..31 of the high one -- leaves group g' = 2 t in the first register and g' = 2 t + 1 in the second, on both
        // halves.  (The residual used to be RE-READ in the epilogue: 77 MB more HBM traffic per launch, PMC 154 vs 77 MB read.)
        f32x16 acc2[RB];
#pragma unroll
        for (int t = 0; t < KC; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(raw[t][e]), __float_as_uint(raw[t][e + 4]), false, false);
                const int ga = 2 * t, gb = 2 * t + 1;
                acc2[ga >> 2][4 * (ga & 3) + e] = __uint_as_float(sw[0]);
                acc2[gb >> 2][4 * (gb & 3) + e] = __uint_as_float(sw[1]);
            }
        if (PREFETCH && tile + stride < p.tiles) load_raw(tile + stride);      // flies under this tile's MFMAs

#pragma unroll 2
        for (int j = 0; j < CH; ++j) {
            f32x16 acc1;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc1[e] = 0.f;
#ifdef MV_I8_PROF
            if (p.dbg & 32) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc1[e] = __uint_as_float(xf[e % KC].x) + j;
            } else
#endif
            {
                uint4 a[KC];                                   // all six fragment reads in flight together
#pragma unroll
                for (int t = 0; t < KC; ++t) a[t] = *(const uint4*)(w1f + j * 32 * W1P + t * 32);
#pragma unroll
                for (int t = 0; t < KC; ++t)
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t]),
                                                                   __builtin_bit_cast(bf16x8, xf[t]), acc1, 0, 0, 0);
            }
            // acc1[4g+i] = h^T[32j + 8g + 4fh + i][my row]: bias + GELU, then registers 8s..8s+7 are the B fragment of k-step s
            uint4 hf[2];
            float4 bq[4];
            uint4 a2[RB][2];                                   // fc2 fragments: issued before the GELU, consumed after it
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[g] = *(const float4*)(b1f + 32 * j + 8 * g);
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) a2[r][s2] = *(const uint4*)(w2f + r * 32 * W2P + (j * 2 + s2) * 32);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#ifdef MV_I8_PROF
                const float4 b = (p.dbg & 4) ? make_float4(0.f, 0.f, 0.f, 0.f) : bq[g];
#else
                const float4 b = bq[g];
#endif
#ifdef MV_I8_PROF
                uint32_t lo, hi;
                if (p.dbg & 1) {
                    lo = pack_bf2(acc1[4 * g] + b.x, acc1[4 * g + 1] + b.y);
                    hi = pack_bf2(acc1[4 * g + 2] + b.z, acc1[4 * g + 3] + b.w);
                } else {
                    lo = pack_bf2(gelu_tanh_f(acc1[4 * g] + b.x), gelu_tanh_f(acc1[4 * g + 1] + b.y));
                    hi = pack_bf2(gelu_tanh_f(acc1[4 * g + 2] + b.z), gelu_tanh_f(acc1[4 * g + 3] + b.w));
                }
#else
                const uint32_t lo = gelu_tanh_pack2(acc1[4 * g] + b.x, acc1[4 * g + 1] + b.y);
                const uint32_t hi = gelu_tanh_pack2(acc1[4 * g + 2] + b.z, acc1[4 * g + 3] + b.w);
#endif
                if (g == 0) { hf[0].x = lo; hf[0].y = hi; }
                if (g == 1) { hf[0].z = lo; hf[0].w = hi; }
                if (g == 2) { hf[1].x = lo; hf[1].y = hi; }
                if (g == 3) { hf[1].z = lo; hf[1].w = hi; }
            }
#ifdef MV_I8_PROF
            if (p.dbg & 2) { acc2[0][0] += __uint_as_float(hf[0].x ^ hf[1].y ^ hf[0].z ^ hf[1].w); continue; }
#endif
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    acc2[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a2[r][s2]),
                                                                      __builtin_bit_cast(bf16x8, hf[s2]), acc2[r], 0, 0, 0);
                }
        }

        // ---- epilogue: acc2[r][4g+i] = y^T[32r + 8g + 4fh + i][my row] (the residual is already inside); 4 consecutive channels per store
        const int m = tile * 32 + fr;
        if (m < p.M) {
            XT* yr = y + (long long)m * C + 4 * fh;
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b = *(const float4*)(b2f + 32 * r + 8 * g);
                    float4 o;
                    o.x = acc2[r][4 * g] + b.x; o.y = acc2[r][4 * g + 1] + b.y; o.z = acc2[r][4 * g + 2] + b.z; o.w = acc2[r][4 * g + 3] + b.w;
#ifdef MV_I8_PROF
                    if ((p.dbg & 16) && o.x != 12345.678f) continue;
#endif
                    Out4<XT>::st(yr + 32 * r + 8 * g, o);
                }
        }
    }
}

int ln_mlp_supported(long long M, int C, int hidden, int x_dtype) {
    return C == lm::C && hidden == lm::H && (x_dtype == MV_F32 || x_dtype == MV_BF16) && M >= 1024 && M < (1ll << 31) - 64 &&
           !get_flag("no_ln_mlp");
}

template <typename XT, int WAVES>
static int ln_mlp_go(const LnMlpP& p, hipStream_t st) {
    int grid = (p.tiles + WAVES - 1) / WAVES;
    if (grid > 256) grid = 256;                              // one block per CU: the weights fill its LDS
    auto kern = ln_mlp96_kernel<XT, WAVES>;
    MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lm::SMEM));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lm::SMEM, st, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int ln_mlp_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, long long M,
                  float eps, int x_dtype, hipStream_t st) {
    LnMlpP p;
    p.x = x; p.w1 = (const bf16_t*)w1; p.b1 = b1; p.w2 = (const bf16_t*)w2; p.b2 = b2; p.y = y;
    p.M = (int)M; p.tiles = (int)((M + 31) / 32); p.eps = eps;
    p.dbg = 0;
#ifdef MV_I8_PROF
    p.dbg = get_flag("ln_mlp_dbg");
#endif
    set_kernel_name(x_dtype == MV_F32 ? "ln_mlp96_f32stream" : "ln_mlp96_bf16stream");
    const int wv = get_flag("ln_mlp_waves");
    if (x_dtype == MV_F32)
        return wv == 16 ? ln_mlp_go<float, 16>(p, st) : wv == 8 ? ln_mlp_go<float, 8>(p, st) : ln_mlp_go<float, 12>(p, st);
    return wv == 16 ? ln_mlp_go<bf16_t, 16>(p, st) : wv == 8 ? ln_mlp_go<bf16_t, 8>(p, st) : ln_mlp_go<bf16_t, 12>(p, st);
}

}  // namespace mv

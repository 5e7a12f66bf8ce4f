// Network entry of the ResNet family in ONE kernel (reference resnet.py:243-254): Conv2d(3 -> 64, 7x7, stride 2,
// pad 3) from the raw NCHW image + folded BatchNorm + ReLU + MaxPool2d(3, stride 2, pad 1) -> NHWC bf16, gfx950.
// The same kernel, other template values, is AlexNet's entry (reference alexnet.py:44-46): Conv2d(3 -> 64, 11x11, stride 4,
// pad 2) + bias + ReLU + MaxPool2d(3, stride 2): the 55 x 55 x 64 map never reaches HBM either (the numbers below are ResNet's;
// AlexNet: 75 x 84 x 3 patch, 66 fragments = 33 k-steps -- the 11-wide filter row is two fragments of 8, the second 3 / 8 full --,
// the weight slab is 69 KB so ONE block per CU, and a wave owns TWO pixel tiles that share every weight fragment it reads).
//
// Un-fused (stem.hip + max-pool) the 112x112x64 map costs a 411 MB write and a 623 MB read per batch of 256
// (profiles/r01 PMC traffic) and 0.38 ms; here it never leaves the CU:
//   * a block owns an 8 x 8 tile of POOLED pixels = 17 x 17 convolution outputs (one halo row / column is
//     recomputed: 289 vs 256 pixels) = a 39 x 39 x 3 input patch, kept in LDS as bf16;
//   * the 289 conv pixels are 10 MFMA pixel tiles of 32 (5 waves x 2); A = the 64 x 176 weight slab in the
//     (c, r, s8) fragment order of stem.hip (zero padded), resident in LDS; B fragments are gathered from the
//     patch (8 consecutive input pixels of one filter row);
//   * scale / shift / ReLU are applied to the fp32 accumulators (per-lane constants for the lane's 32
//     channels), the bf16 result goes to an LDS tile [289][64]; conv positions outside the image store 0, which
//     is neutral for a max over post-ReLU values (every pool window holds at least one real pixel);
//   * after a barrier every thread pools 8 channels of one output pixel: 9 x ds_read_b128 + packed int16 max
//     (non-negative bf16 order like their bit patterns) and ONE 16-byte store, 8 lanes per 128-byte NHWC line;
//   * TWO blocks share a CU (<= 128 VGPRs: `amdgpu_waves_per_eu(4, 4)` -- with 154 registers and five waves per block
//     the second block was never co-resident, measured with per-block time stamps), so one block's patch load and
//     pooling run under the other's MFMAs; the patch is fetched as aligned 4-pixel chunks (columns 4 px0 - 8 ..),
//     5 loads per thread instead of 15 scalar ones.
#include "mfma_common.h"

namespace mv {

struct StemPoolP {
    const void* x;
    const bf16_t* w;        // OIHW [64][3][7][7]
    const float* scale;
    const float* shift;
    bf16_t* y;              // [N][Po][Qo][64]
    int N, H, W, Ho, Wo, Po, Qo;
    int tiles_y, tiles_x, tiles;
};

template <typename TX> __device__ __forceinline__ float ld_img(const TX* p);
template <> __device__ __forceinline__ float ld_img<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_img<bf16_t>(const bf16_t* p) { return bf2f(*p); }

typedef short i16x8 __attribute__((ext_vector_type(8)));   // post-ReLU bf16 orders like int16 (and -0.0 = 0x8000 never wins)

template <typename TX> struct Img4;
template <> struct Img4<float> {
    static __device__ __forceinline__ void ld(const float* p, float* v) {
        const float4 t = *(const float4*)p;
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
};
template <> struct Img4<bf16_t> {
    static __device__ __forceinline__ void ld(const bf16_t* p, float* v) {
        const uint2 t = *(const uint2*)p;
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
};

// VEC: W % 4 == 0 and a 16-byte aligned image base -- every 4-pixel chunk is one aligned load, entirely inside or
// entirely outside the image; otherwise the chunk is assembled from four bounds-checked scalar loads.
// R = S: filter edge; STRIDE, PAD: of the convolution; PPAD: of the 3 x 3 / 2 max-pool; PWP: patch row pitch (elements, multiple of 4);
// TPW: pixel tiles a wave works on at once (they share the weight fragments).  Columns: the patch starts at image column
// 2 STRIDE px0 - ALN (a multiple of 4: aligned 4-pixel chunks), the window of conv column cx at patch column STRIDE cx + WO; a B
// fragment must start on an even element (4-byte aligned ds_read_b32): SHIFT = 1 starts it one pixel LEFT of the window.
template <typename TX, bool VEC, int R, int STRIDE, int PAD, int PPAD, int PWP, int TPW>
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(4, 4))) void stem_pool_kernel(const StemPoolP p) {
    constexpr int C = 3, S = R, K = 64;
    constexpr int CT = 17;                                  // conv tile edge (2 * 8 + 1)
    constexpr int NPIX = CT * CT;                           // 289
    constexpr int BC = STRIDE * PPAD + PAD;                 // image row / column of conv (2 py0, 2 px0)'s window = 2 STRIDE py0 - BC
    constexpr int ALN = (BC + 3) & ~3, WO = ALN - BC, SHIFT = WO & 1;
    constexpr int SB = (S + SHIFT + 7) / 8;                 // fragments of 8 per filter row
    constexpr int PH = STRIDE * (CT - 1) + R;               // 39 patch rows
    constexpr int PWp = PWP;                                // patch row pitch (elements): image columns 4 px0 - 8 .. + 47
    static_assert(PWp % 4 == 0 && PWp >= STRIDE * (CT - 1) + WO - SHIFT + 8 * SB, "patch row must hold the last fragment");
    constexpr int PATCH = C * PH * PWp;                     // 5616
    constexpr int NCHUNK = C * PH * (PWp / 4);              // 1404 chunks of 4 pixels
    constexpr int NFRAG = C * R * SB;                       // 21 fragments of 8 (s padded 7 -> 8)
    constexpr int NK16 = (NFRAG + 1) / 2;                   // 11
    constexpr int WPITCH = ((2 * NK16) | 1) * 16;           // 368 bytes
    constexpr int CPITCH = 144;                             // conv tile row pitch (bytes): 64 bf16 + 16
    constexpr int NT = 320, NEMAX = 8, NE = (NCHUNK + NT - 1) / NT;    // 5 chunks per thread
    constexpr int NPASS = (NE + NEMAX - 1) / NEMAX, NEP = (NE + NPASS - 1) / NPASS;   // patch load in passes of <= 8 chunks per thread
    constexpr int OFF_PATCH = K * WPITCH;                                  // 23552
    constexpr int OFF_CTILE = OFF_PATCH + ((PATCH * 2 + 15) & ~15);         // + 11232
    constexpr int OFF_FTAB = OFF_CTILE + NPIX * CPITCH;                     // + 41616
    static_assert(K * C * R * S * 2 <= OFF_FTAB - OFF_PATCH, "raw weights are staged in the patch + conv tile area");
    extern __shared__ __attribute__((aligned(16))) char smem[];             // 76.9 KB: two blocks per CU
    char* wl = smem;
    bf16_t* patch = (bf16_t*)(smem + OFF_PATCH);
    char* ctile = smem + OFF_CTILE;
    float* sct = (float*)(smem + OFF_FTAB);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;

    // weight slab (k' = (c, r, s8) order, s8 = [0, s0 .. s6]: a B fragment starts one pixel LEFT of the window so that
    // it is 4-byte aligned in the patch), once per block: the 18.8 KB of OIHW weights are copied to LDS as they are
    // (coalesced 16-byte loads, all in flight; the conv tile area is idle) and re-laid out LDS -> LDS.  Seven dependent
    // 2-byte global loads per fragment made this prologue ~5 us long.
    {
        constexpr int WELEMS = K * C * R * S;                  // 9408 bf16
        constexpr int NCH16 = (WELEMS * 2 + 15) / 16;          // 1176 chunks
        constexpr int U = (NCH16 + NT - 1) / NT;               // 4
        bf16_t* wraw = (bf16_t*)(smem + OFF_PATCH);        // patch + conv tile area, idle now
        const bool al = ((uintptr_t)p.w & 15) == 0;
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = u * NT + tid;
            const int ic = i < NCH16 ? i : NCH16 - 1;
            if (al) v[u] = *(const uint4*)(p.w + ic * 8);
        }
        if (al) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = u * NT + tid;
                if (i < NCH16) *(uint4*)(wraw + i * 8) = v[u];
            }
        } else {
            for (int i = tid; i < WELEMS; i += NT) wraw[i] = p.w[i];
        }
        __syncthreads();
        for (int i = tid; i < K * 2 * NK16; i += NT) {
            const int row = i / (2 * NK16), f = i - row * (2 * NK16);
            uint32_t u[4] = {0, 0, 0, 0};
            if (f < NFRAG) {
                const int cr = f / SB, sb = f - cr * SB;
                const bf16_t* src = wraw + (row * C * R + cr) * S;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int sx = 8 * sb + e - SHIFT;           // SHIFT = 1: slot 0 = pad
                    if (sx >= 0 && sx < S) u[e >> 1] |= (uint32_t)src[sx] << ((e & 1) * 16);
                }
            }
            *(uint4*)(wl + row * WPITCH + f * 16) = make_uint4(u[0], u[1], u[2], u[3]);
        }
    }

    // epilogue constants live in LDS (64 + 64 floats): as per-lane registers they cost 64 VGPRs and the second
    // block per CU; accumulator quad g of channel tile a holds channels a*32 + 8g + 4fh + 0..3
    if (tid < 64) {
        sct[tid] = p.scale ? p.scale[tid] : 1.f;
        sct[64 + tid] = p.shift ? p.shift[tid] : 0.f;
    }

    const TX* xg = (const TX*)p.x;
    const int HW = p.H * p.W;
    auto origin = [&](int tile, int& b, int& py0, int& px0) {
        const int tx = tile % p.tiles_x;
        const int ty = (tile / p.tiles_x) % p.tiles_y;
        b = tile / (p.tiles_x * p.tiles_y);
        py0 = ty * 8;
        px0 = tx * 8;
    };
    // chunk i of thread: i = j * NT + tid -> (c, yy, q): input row 4*py0 - 5 + yy, columns 4*px0 - 8 + 4q .. + 3
    // XCD-aware tile order: blocks are dispatched round-robin over the 8 XCDs, each with its own L2.  XCD x owns the
    // contiguous tile range [x T/8, (x+1) T/8) and its blocks stride through it together, so the overlapping halo
    // columns / rows of neighbouring tiles are fetched from HBM once per XCD instead of once per tile (PMC: 224 MB
    // fetched per 128 images for 77 MB of pixels with the plain grid-stride order).
    const bool by_xcd = gridDim.x >= 8;
    const int xcd = by_xcd ? (blockIdx.x & 7) : 0, jloc = by_xcd ? (blockIdx.x >> 3) : blockIdx.x;
    const int per_x = by_xcd ? ((gridDim.x + 7 - xcd) >> 3) : gridDim.x;        // blocks that landed on my XCD
    const int t_lo = by_xcd ? (int)(((long long)p.tiles * xcd) >> 3) : 0;
    const int t_hi = by_xcd ? (int)(((long long)p.tiles * (xcd + 1)) >> 3) : p.tiles;
    for (int tile = t_lo + jloc; tile < t_hi; tile += per_x) {
        int b, py0, px0;
        origin(tile, b, py0, px0);
        const int hi0 = 2 * STRIDE * py0 - BC, wi0 = 2 * STRIDE * px0 - ALN;
        const TX* xb = xg + (long long)b * C * HW;
#pragma unroll 1
        for (int ps = 0; ps < NPASS; ++ps) {
            float pv[NEP][4];
#pragma unroll
            for (int j = 0; j < NEP; ++j) {
                const int i = (ps * NEP + j) * NT + tid;
                const int q = i % (PWp / 4), t2 = i / (PWp / 4);
                const int yy = t2 % PH, c = t2 / PH;
                const int hi = hi0 + yy, wi = wi0 + 4 * q;
                const bool rowok = i < NCHUNK && (unsigned)hi < (unsigned)p.H;
                if constexpr (VEC) {
                    const bool ok = rowok && (unsigned)wi < (unsigned)p.W;
                    Img4<TX>::ld(xb + (ok ? (long long)c * HW + (long long)hi * p.W + wi : 0), pv[j]);
                    if (!ok) pv[j][0] = pv[j][1] = pv[j][2] = pv[j][3] = 0.f;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool ok = rowok && (unsigned)(wi + e) < (unsigned)p.W;
                        const float t = ld_img<TX>(xb + (ok ? (long long)c * HW + (long long)hi * p.W + wi + e : 0));
                        pv[j][e] = ok ? t : 0.f;
                    }
                }
            }
            if (ps == 0) __syncthreads();         // previous tile: MFMAs done with the patch, pooling done with ctile (the loads above fly across)
#pragma unroll
            for (int j = 0; j < NEP; ++j) {
                const int i = (ps * NEP + j) * NT + tid;
                if (i < NCHUNK) {
                    uint2 u;
                    u.x = pack_bf2(pv[j][0], pv[j][1]);
                    u.y = pack_bf2(pv[j][2], pv[j][3]);
                    *(uint2*)(patch + 4 * i) = u;
                }
            }
        }
        __syncthreads();

        // ---- convolution: 10 pixel tiles of 32 over the 17 x 17 conv region (conv origin 2*py0 - PPAD, 2*px0 - PPAD), TPW at a time
        constexpr int NTILE = (NPIX + 31) / 32;
#pragma unroll 1
        for (int t0 = wave * TPW; t0 < NTILE; t0 += 5 * TPW) {
            int idx[TPW], lbase[TPW], cyx[TPW];
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                idx[u] = (t0 + u) * 32 + fr;
                const int ic = idx[u] < NPIX ? idx[u] : NPIX - 1;
                const int cy = ic / CT, cx = ic - cy * CT;
                cyx[u] = cy * 32 + cx;
                lbase[u] = (STRIDE * cy) * PWp + STRIDE * cx + WO - SHIFT;   // SHIFT = 1: one pixel left of my window origin -> even
            }
            f32x16 acc[TPW][2];
#pragma unroll
            for (int u = 0; u < TPW; ++u)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[u][a][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < NK16; ++kk) {
                const int f = 2 * kk + fh;
                // fragment f = ((c, r), sb) starts at patch element (c*PH + r)*PWp + 8 sb: two compile-time constants selected by
                // the lane half (an LDS table here put TWO dependent LDS round trips in front of every MFMA pair)
                const int f0 = 2 * kk < NFRAG ? 2 * kk : NFRAG - 1, f1 = 2 * kk + 1 < NFRAG ? 2 * kk + 1 : NFRAG - 1;
                const int o0 = (((f0 / SB) / R) * PH + (f0 / SB) % R) * PWp + 8 * (f0 % SB);
                const int o1 = (((f1 / SB) / R) * PH + (f1 / SB) % R) * PWp + 8 * (f1 % SB);
                const uint4 a0 = *(const uint4*)(wl + fr * WPITCH + f * 16);
                const uint4 a1 = *(const uint4*)(wl + (32 + fr) * WPITCH + f * 16);
#pragma unroll
                for (int u = 0; u < TPW; ++u) {
                    const uint32_t* src = (const uint32_t*)(patch + lbase[u] + (fh ? o1 : o0));   // 4-byte aligned
                    const uint4 bv = make_uint4(src[0], src[1], src[2], src[3]);
                    acc[u][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0),
                                                                        __builtin_bit_cast(bf16x8, bv), acc[u][0], 0, 0, 0);
                    acc[u][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1),
                                                                        __builtin_bit_cast(bf16x8, bv), acc[u][1], 0, 0, 0);
                }
            }
            // BN + ReLU in fp32, one bf16 rounding, 8-byte stores into the conv tile; 0 outside the conv map
#pragma unroll
            for (int u = 0; u < TPW; ++u) {
                const int oy = 2 * py0 - PPAD + (cyx[u] >> 5), ox = 2 * px0 - PPAD + (cyx[u] & 31);
                const bool inside = (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
                if (t0 + u < NTILE && idx[u] < NPIX) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = a * 32 + 8 * g + 4 * fh;
                            const float4 s4 = *(const float4*)(sct + n), h4 = *(const float4*)(sct + 64 + n);
                            const float scv[4] = {s4.x, s4.y, s4.z, s4.w}, shv[4] = {h4.x, h4.y, h4.z, h4.w};
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = fmaxf(fmaf(acc[u][a][4 * g + e], scv[e], shv[e]), 0.f);
                                v[e] = inside ? v[e] : 0.f;
                            }
                            uint2 w2;
                            w2.x = pack_bf2(v[0], v[1]);
                            w2.y = pack_bf2(v[2], v[3]);
                            *(uint2*)(ctile + idx[u] * CPITCH + (a * 32 + 8 * g + 4 * fh) * 2) = w2;
                        }
                }
            }
        }
        __syncthreads();

        // ---- max-pool 3x3 / 2 over the conv tile: item = (pooled pixel 0..63, 8-channel group 0..7)
        for (int it = tid; it < 64 * 8; it += NT) {
            const int c8 = it & 7, pp = it >> 3;
            const int ly = pp >> 3, lx = pp & 7;
            const int py = py0 + ly, px = px0 + lx;
            const char* base = ctile + ((2 * ly) * CT + 2 * lx) * CPITCH + c8 * 16;   // window origin = conv (2ly, 2lx) of the tile
            i16x8 m = *(const i16x8*)base;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if (dy == 0 && dx == 0) continue;
                    const i16x8 v = *(const i16x8*)(base + (dy * CT + dx) * CPITCH);
                    m = __builtin_elementwise_max(m, v);
                }
            if (py < p.Po && px < p.Qo)
                *(i16x8*)(p.y + (((long long)b * p.Po + py) * p.Qo + px) * K + c8 * 8) = m;
        }
    }
}

// the two entries this kernel is built for: ResNet (7x7 / 2, pad 3, pool pad 1) and AlexNet (11x11 / 4, pad 2, pool pad 0)
int stem_pool_supported(int C, int K, int R, int S, int sh, int sw, int ph, int pw, int pk, int ps, int pp, int act,
                        int x_dtype, int out_dtype, long long in_elems) {
    const bool resnet = R == 7 && sh == 2 && ph == 3 && pp == 1;
    const bool alexnet = R == 11 && sh == 4 && ph == 2 && pp == 0 && !get_flag("no_stem_pool11");
    return C == 3 && K == 64 && R == S && sh == sw && ph == pw && pk == 3 && ps == 2 && (resnet || alexnet) &&
           act == MV_ACT_RELU && (x_dtype == MV_F32 || x_dtype == MV_BF16) && out_dtype == MV_BF16 && in_elems < (1LL << 31);
}

template <int R, int STRIDE, int PAD, int PPAD, int PWP, int TPW>
static int stem_pool_go(StemPoolP p, int x_dtype, int blocks_per_cu, const char* name_f32, const char* name_bf16, hipStream_t st) {
    constexpr int SB = (R + (((((STRIDE * PPAD + PAD) + 3) & ~3) - (STRIDE * PPAD + PAD)) & 1) + 7) / 8;
    constexpr int NK16 = (3 * R * SB + 1) / 2, WPITCH = ((2 * NK16) | 1) * 16, PH = STRIDE * 16 + R;
    constexpr int SMEM = 64 * WPITCH + ((3 * PH * PWP * 2 + 15) & ~15) + 289 * 144 + 128 * 4;
    static_assert(SMEM <= 160 * 1024, "LDS");
    p.Ho = (p.H + 2 * PAD - R) / STRIDE + 1;
    p.Wo = (p.W + 2 * PAD - R) / STRIDE + 1;
    p.Po = (p.Ho + 2 * PPAD - 3) / 2 + 1;
    p.Qo = (p.Wo + 2 * PPAD - 3) / 2 + 1;
    p.tiles_y = (p.Po + 7) / 8;
    p.tiles_x = (p.Qo + 7) / 8;
    const long long tiles = (long long)p.N * p.tiles_y * p.tiles_x;
    if (tiles >= (1LL << 31)) {
        set_error("stem_pool: too many tiles");
        return MV_E_UNSUPPORTED;
    }
    p.tiles = (int)tiles;
    const int cap = 256 * blocks_per_cu;                  // persistent blocks
    const int gx = p.tiles < cap ? p.tiles : cap;
    set_kernel_name(x_dtype == MV_F32 ? name_f32 : name_bf16);
    const bool vec = p.W % 4 == 0 && ((uintptr_t)p.x & 15) == 0;
#define GO(TX_, V_)                                                                                              \
    do {                                                                                                         \
        auto kern = stem_pool_kernel<TX_, V_, R, STRIDE, PAD, PPAD, PWP, TPW>;                                   \
        MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));        \
        hipLaunchKernelGGL(kern, dim3(gx), dim3(320), SMEM, st, p);                                              \
    } while (0)
    if (x_dtype == MV_F32) { if (vec) GO(float, true); else GO(float, false); }
    else { if (vec) GO(bf16_t, true); else GO(bf16_t, false); }
#undef GO
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int stem_pool_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int H, int W,
                     int R, int x_dtype, hipStream_t st) {
    StemPoolP p;
    p.x = x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.y = (bf16_t*)y;
    p.N = N; p.H = H; p.W = W;
    if (R == 11)                                          // 145 KB of LDS: one block per CU; two pixel tiles per wave at once (+1.5 % on the model;
                                                          // the same on the ResNet entry is 2.7 % SLOWER: it costs the second block per CU its slack)
        return stem_pool_go<11, 4, 2, 0, 84, 2>(p, x_dtype, 1, "stem_pool11_mfma_f32in", "stem_pool11_mfma_bf16in", st);
    return stem_pool_go<7, 2, 3, 1, 48, 1>(p, x_dtype, 2, "stem_pool_mfma_f32in", "stem_pool_mfma_bf16in", st);   // two blocks per CU
}

}  // namespace mv

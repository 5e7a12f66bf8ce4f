// Network entry of the ResNet family in ONE kernel (reference resnet.py:243-254): Conv2d(3 -> 64, 7x7, stride 2,
// pad 3) from the raw NCHW image + folded BatchNorm + ReLU + MaxPool2d(3, stride 2, pad 1) -> NHWC bf16, gfx950.
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
template <typename TX, bool VEC>
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(4, 4))) void stem_pool_kernel(const StemPoolP p) {
    constexpr int C = 3, R = 7, S = 7, K = 64;
    constexpr int CT = 17;                                  // conv tile edge (2 * 8 + 1)
    constexpr int NPIX = CT * CT;                           // 289
    constexpr int PH = 2 * (CT - 1) + R;                    // 39 patch rows
    constexpr int PWp = 48;                                 // patch row pitch (elements): image columns 4 px0 - 8 .. + 47
    constexpr int PATCH = C * PH * PWp;                     // 5616
    constexpr int NCHUNK = C * PH * (PWp / 4);              // 1404 chunks of 4 pixels
    constexpr int NFRAG = C * R;                            // 21 fragments of 8 (s padded 7 -> 8)
    constexpr int NK16 = (NFRAG + 1) / 2;                   // 11
    constexpr int WPITCH = ((2 * NK16) | 1) * 16;           // 368 bytes
    constexpr int CPITCH = 144;                             // conv tile row pitch (bytes): 64 bf16 + 16
    constexpr int NT = 320, NE = (NCHUNK + NT - 1) / NT;    // 5 chunks per thread
    constexpr int OFF_PATCH = K * WPITCH;                                  // 23552
    constexpr int OFF_CTILE = OFF_PATCH + ((PATCH * 2 + 15) & ~15);         // + 11232
    constexpr int OFF_FTAB = OFF_CTILE + NPIX * CPITCH;                     // + 41616
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
        bf16_t* wraw = (bf16_t*)ctile;
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
                const bf16_t* src = wraw + (row * C * R + f) * S;
#pragma unroll
                for (int e = 0; e < S; ++e) u[(e + 1) >> 1] |= (uint32_t)src[e] << (((e + 1) & 1) * 16);   // slot 0 = pad
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
        const int hi0 = 4 * py0 - 5, wi0 = 4 * px0 - 8;
        const TX* xb = xg + (long long)b * C * HW;
        float pv[NE][4];
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int i = j * NT + tid;
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
        __syncthreads();                      // previous tile: MFMAs done with the patch, pooling done with ctile
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int i = j * NT + tid;
            if (i < NCHUNK) {
                uint2 u;
                u.x = pack_bf2(pv[j][0], pv[j][1]);
                u.y = pack_bf2(pv[j][2], pv[j][3]);
                *(uint2*)(patch + 4 * i) = u;
            }
        }
        __syncthreads();

        // ---- convolution: 10 pixel tiles of 32 over the 17 x 17 conv region (conv origin 2*py0 - 1, 2*px0 - 1)
#pragma unroll 1
        for (int t = wave; t < (NPIX + 31) / 32; t += 5) {
            const int idx = t * 32 + fr;
            const int ic = idx < NPIX ? idx : NPIX - 1;
            const int cy = ic / CT, cx = ic - cy * CT;
            const int lbase = (2 * cy) * PWp + 2 * cx + 2;  // one pixel left of my window origin (column 3 + 2cx): even
            f32x16 acc[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
#pragma unroll
            for (int kk = 0; kk < NK16; ++kk) {
                const int f = 2 * kk + fh;
                // fragment f = (c, r) starts at patch element (c*PH + r)*PWp: two compile-time constants selected by
                // the lane half (an LDS table here put TWO dependent LDS round trips in front of every MFMA pair)
                const int f0 = 2 * kk < NFRAG ? 2 * kk : NFRAG - 1, f1 = 2 * kk + 1 < NFRAG ? 2 * kk + 1 : NFRAG - 1;
                const int o0 = ((f0 / R) * PH + f0 % R) * PWp, o1 = ((f1 / R) * PH + f1 % R) * PWp;
                const uint32_t* src = (const uint32_t*)(patch + lbase + (fh ? o1 : o0));   // 4-byte aligned
                const uint4 bv = make_uint4(src[0], src[1], src[2], src[3]);
                const uint4 a0 = *(const uint4*)(wl + fr * WPITCH + f * 16);
                const uint4 a1 = *(const uint4*)(wl + (32 + fr) * WPITCH + f * 16);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0),
                                                                 __builtin_bit_cast(bf16x8, bv), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1),
                                                                 __builtin_bit_cast(bf16x8, bv), acc[1], 0, 0, 0);
            }
            // BN + ReLU in fp32, one bf16 rounding, 8-byte stores into the conv tile; 0 outside the conv map
            const int oy = 2 * py0 - 1 + cy, ox = 2 * px0 - 1 + cx;
            const bool inside = (unsigned)oy < (unsigned)p.Ho && (unsigned)ox < (unsigned)p.Wo;
            if (idx < NPIX) {
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
                            v[e] = fmaxf(fmaf(acc[a][4 * g + e], scv[e], shv[e]), 0.f);
                            v[e] = inside ? v[e] : 0.f;
                        }
                        uint2 u;
                        u.x = pack_bf2(v[0], v[1]);
                        u.y = pack_bf2(v[2], v[3]);
                        *(uint2*)(ctile + idx * CPITCH + (a * 32 + 8 * g + 4 * fh) * 2) = u;
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

int stem_pool_supported(int C, int K, int R, int S, int sh, int sw, int ph, int pw, int pk, int ps, int pp, int act,
                        int x_dtype, int out_dtype, long long in_elems) {
    return C == 3 && K == 64 && R == 7 && S == 7 && sh == 2 && sw == 2 && ph == 3 && pw == 3 && pk == 3 && ps == 2 && pp == 1 &&
           act == MV_ACT_RELU && (x_dtype == MV_F32 || x_dtype == MV_BF16) && out_dtype == MV_BF16 && in_elems < (1LL << 31);
}

int stem_pool_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int H, int W,
                     int x_dtype, hipStream_t st) {
    StemPoolP p;
    p.x = x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.y = (bf16_t*)y;
    p.N = N; p.H = H; p.W = W;
    p.Ho = (H + 6 - 7) / 2 + 1;
    p.Wo = (W + 6 - 7) / 2 + 1;
    p.Po = (p.Ho + 2 - 3) / 2 + 1;
    p.Qo = (p.Wo + 2 - 3) / 2 + 1;
    p.tiles_y = (p.Po + 7) / 8;
    p.tiles_x = (p.Qo + 7) / 8;
    const long long tiles = (long long)N * p.tiles_y * p.tiles_x;
    if (tiles >= (1LL << 31)) {
        set_error("stem_pool: too many tiles");
        return MV_E_UNSUPPORTED;
    }
    p.tiles = (int)tiles;
    int gx = p.tiles < 512 ? p.tiles : 512;               // two persistent blocks per CU
    set_kernel_name(x_dtype == MV_F32 ? "stem_pool_mfma_f32in" : "stem_pool_mfma_bf16in");
    constexpr int SMEM = 64 * 368 + 11232 + 289 * 144 + 128 * 4;
    const bool vec = W % 4 == 0 && ((uintptr_t)x & 15) == 0;
#define GO(TX_, V_)                                                                                              \
    do {                                                                                                         \
        auto kern = stem_pool_kernel<TX_, V_>;                                                                   \
        MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));        \
        hipLaunchKernelGGL(kern, dim3(gx), dim3(320), SMEM, st, p);                                              \
    } while (0)
    if (x_dtype == MV_F32) { if (vec) GO(float, true); else GO(float, false); }
    else { if (vec) GO(bf16_t, true); else GO(bf16_t, false); }
#undef GO
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // namespace mv

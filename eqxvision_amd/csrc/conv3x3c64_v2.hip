// 3x3 stride-1 convolution 64 -> 64 channels (ResNet-50 layer1 conv2, 56x56), second design, gfx950.
//
// conv3x3c64.hip fetches every MFMA B fragment (8 channels of one input pixel of one tap) straight from
// global memory: 32 different 128-byte lines per load instruction, 32 bytes used of each, and every line is
// requested 9 x 4 times -- the L1 tag path, not HBM or the matrix pipe, bounds it (114 us, 0.52 PFLOP/s).
// Here
//   * the WEIGHTS are register-stationary: a wave owns 32 output channels and keeps all 36 A fragments of
//     them (9 taps x 4 k16-steps x 16 bytes per lane = 144 VGPRs) for the whole kernel -- no LDS, no re-reads;
//   * the INPUT is staged once per tile through LDS: a tile is a band of 4 output rows x up to 62 columns, its
//     6 x (cols+2) halo (row pitch 64 pixels) goes in by LDS-DMA as whole 128-byte pixel rows (full lines, each read once per tile),
//     double buffered one tile ahead, XOR-swizzled on the source address so the fragment reads
//     (`ds_read_b128`, inline asm) are conflict-free; out-of-image halo pixels read the zero page;
//   * 8 waves = 4 pixel groups x 2 channel halves; a pixel tile of 32 is 36 MFMAs per wave;
//   * epilogue as everywhere: wave-private LDS transpose, fp32 scale/shift/act, 16-byte stores.
#include "igemm_pipe.h"

namespace mv {

// LDS accesses of the steady state are inline asm: a compiler-visible LDS read or write after an LDS-DMA makes hipcc
// wait vmcnt(0) first -- here that would park every epilogue behind the NEXT tile's halo DMA.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_write16(unsigned addr, float a, float b, float c, float d) {
    const f32x4v v = {a, b, c, d};
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

struct C3V2P {
    const bf16_t* x;
    const bf16_t* w;      // KRSC [64][3][3][64]
    const float* scale;
    const float* shift;
    bf16_t* y;
    const bf16_t* zero;
    int N, H, W, act;
    int tiles_y, tiles_x, tiles, tc;      // bands of 4 rows, column tiles of `tc` (<= 64) columns
};

__global__ __launch_bounds__(512) void conv3x3c64_v2_kernel(const C3V2P p) {
    constexpr int TR = 4, HR = TR + 2;
    constexpr int HC = 64;                                  // halo row pitch in pixels (column tiles are <= 62 wide): with a
                                                            // multiple of 16 the swizzle term (hp >> 1) & 7 depends on the halo
                                                            // COLUMN only, so a tap's row offset is a plain scalar add
    constexpr int HBUF = HR * HC * 128;                     // one halo buffer: 48 KB
    constexpr int EPITCH = 32 * 4 + 16;                     // epilogue patch row: 32 channels fp32 + pad
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = wave & 1, pg = wave >> 1;              // channel half, pixel group
    const int fr = lane & 31, fh = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

    // ---- my 32 output channels' weights: A fragment of k16-step (tap, kk) = w[ch][tap][kk*16 + fh*8 ..+8].
    // The block fetches the 72 KB once, coalesced, into LDS (the second halo buffer and the patches are idle until
    // the main loop) and every lane picks its 36 fragments from there: read straight from global, the 36 strided
    // loads per lane of all 256 CUs hit the same lines 16 times over (7.3 us of prologue by time stamps; 2 us now).
    uint4 wreg[36];
    {
        char* wst = smem + HBUF;                               // [64 rows][1152 + 16 bytes]
        constexpr int WROW = 576 * 2 + 16;
        static_assert(HBUF + 64 * WROW <= 2 * HBUF + 8 * 32 * EPITCH, "weight staging must stay below the sct table");
        uint4 wv[9];                                           // 64 rows x 72 chunks of 16 bytes = 9 per thread, all in flight
#pragma unroll
        for (int j = 0; j < 9; ++j) wv[j] = *(const uint4*)(p.w + (long long)(j * 512 + tid) * 8);
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int i = j * 512 + tid, row = i / 72, ch = i - row * 72;
            *(uint4*)(wst + row * WROW + ch * 16) = wv[j];
        }
        __syncthreads();
        const char* wr = wst + (half * 32 + fr) * WROW + fh * 16;
#pragma unroll
        for (int ks = 0; ks < 36; ++ks) wreg[ks] = *(const uint4*)(wr + (ks >> 2) * 128 + (ks & 3) * 32);
    }
    // scale / shift of the 64 channels in LDS (read per pixel tile: as registers they pushed the kernel into spills)
    float* sct = (float*)(smem + 2 * HBUF + 8 * 32 * EPITCH);
    if (tid < 64) {
        sct[tid] = p.scale ? p.scale[tid] : 1.f;
        sct[64 + tid] = p.shift ? p.shift[tid] : 0.f;
    }
    __syncthreads();
    const unsigned sct_a = lds0 + 2 * HBUF + 8 * 32 * EPITCH + (half * 32 + (lane & 3) * 8) * 4;
    const unsigned ep_a = lds0 + 2 * HBUF + wave * (32 * EPITCH);

    auto tile_org = [&](int t, int& b, int& r0, int& c0, int& tcw) {
        const int tx = t % p.tiles_x;
        const int ty = (t / p.tiles_x) % p.tiles_y;
        b = t / (p.tiles_x * p.tiles_y);
        r0 = ty * TR;
        c0 = tx * p.tc;
        tcw = (p.W - c0) < p.tc ? (p.W - c0) : p.tc;
    };
    // LDS-DMA of the halo of tile t: 48 pieces of 8 halo pixels (1 KB), 6 per wave; halo pixel hp = hy * 64 + hx
    auto stage = [&](int t, int buf) {
        int b, r0, c0, tcw;
        tile_org(t, b, r0, c0, tcw);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int pc = wave + 8 * j;
            const int hp = pc * 8 + (lane >> 3);
            const int hy = hp >> 6, hx = hp & 63;
            const int iy = r0 - 1 + hy, ix = c0 - 1 + hx;
            const int chunk = (lane & 7) ^ ((hx >> 1) & 7);
            const bool ok = hx < tcw + 2 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const bf16_t* src = ok ? p.x + (((long long)b * p.H + iy) * p.W + ix) * 64 + chunk * 8 : p.zero;
            glds16(src, smem + buf * HBUF + pc * 1024);
        }
    };

    int t = blockIdx.x;
    if (t < p.tiles) stage(t, 0);
    int cur = 0;
    for (; t < p.tiles; t += gridDim.x) {
        int b, r0, c0, tcw;
        tile_org(t, b, r0, c0, tcw);
        const int npx = TR * tcw;
        wait_vm<0>();                                          // my pieces of this tile's halo have landed
        __builtin_amdgcn_s_barrier();                          // ... everybody's; and the other buffer is free
        if (t + (int)gridDim.x < p.tiles) stage(t + gridDim.x, cur ^ 1);
        const unsigned hb = lds0 + cur * HBUF;

        for (int pt = pg; pt * 32 < npx; pt += 4) {
            const int pi = pt * 32 + fr;
            const int pic = pi < npx ? pi : npx - 1;
            const int py = pic / tcw, px = pic - py * tcw;
            // fragment addresses: tap (r, s), k16-step kk, lane half fh -> halo pixel (py + r, px + s), 16-byte slot
            // (2kk + fh) ^ (((px + s) >> 1) & 7): a per-lane base and swizzle per s, the row r is an immediate offset
            unsigned pbase[3], sw16[3];
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) {
                const unsigned hx = (unsigned)(px + sx);
                pbase[sx] = hb + ((unsigned)(py * HC) + hx) * 128u;
                sw16[sx] = ((hx >> 1) & 7u) << 4;
            }
            f32x16 acc;
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[e] = 0.f;
            u32x4 bf[2][4];     // B fragments one tap ahead.  (Two taps ahead -- a tap is only 4 MFMAs = 128 cycles of matrix work -- was
                                // built in round 6 and is 7 % SLOWER, 44.8 vs 41.8 us alone, resnet50 -0.9 %: 256 instead of 247 VGPRs
                                // and three taps of reads in the LDS queue; profiles/r06/conv3x3c64_prefetch_two_taps_ab.txt)
            auto read_tap = [&](int set, int tap) {
                const int r = tap / 3, sx = tap - 3 * r;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const unsigned a = pbase[sx] + ((((unsigned)(2 * kk + fh)) << 4) ^ sw16[sx]);   // one v_xad_u32
                    if (r == 0) lds_read16<0>(bf[set][kk], a);
                    else if (r == 1) lds_read16<HC * 128>(bf[set][kk], a);
                    else lds_read16<2 * HC * 128>(bf[set][kk], a);
                }
            };
            read_tap(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int cs = tap & 1;
                if (tap < 8) read_tap(cs ^ 1, tap + 1);
                if (tap < 8) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bf[cs][0]), "+v"(bf[cs][1]), "+v"(bf[cs][2]), "+v"(bf[cs][3]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bf[cs][0]), "+v"(bf[cs][1]), "+v"(bf[cs][2]), "+v"(bf[cs][3]));
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[tap * 4 + kk]),
                                                                  __builtin_bit_cast(bf16x8, bf[cs][kk]), acc, 0, 0, 0);
            }
            // ---- epilogue: [32 px][32 ch] fp32 patch -> row-major, 4 lanes x 16 bytes = my 64-byte half line.
            // The hazard recogniser does not look into inline asm: the last MFMA's result needs 18 wait states
            // before a DS instruction may read it (without them two parity cases failed).
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(acc));
#pragma unroll
            for (int g = 0; g < 4; ++g)
                lds_write16(ep_a + fr * EPITCH + (8 * g + 4 * fh) * 4, acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            u32x4 s0, s1, h0, h1, lo[2], hi[2];
            lds_read16<0>(s0, sct_a);
            lds_read16<16>(s1, sct_a);
            lds_read16<256>(h0, sct_a);
            lds_read16<272>(h1, sct_a);
            const int c4 = lane & 3;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const unsigned ra = ep_a + (pass * 16 + (lane >> 2)) * EPITCH + c4 * 32;
                lds_read16<0>(lo[pass], ra);
                lds_read16<16>(hi[pass], ra);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(h0), "+v"(h1), "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]));
            const float scv[8] = {__uint_as_float(s0.x), __uint_as_float(s0.y), __uint_as_float(s0.z), __uint_as_float(s0.w),
                                  __uint_as_float(s1.x), __uint_as_float(s1.y), __uint_as_float(s1.z), __uint_as_float(s1.w)};
            const float shv[8] = {__uint_as_float(h0.x), __uint_as_float(h0.y), __uint_as_float(h0.z), __uint_as_float(h0.w),
                                  __uint_as_float(h1.x), __uint_as_float(h1.y), __uint_as_float(h1.z), __uint_as_float(h1.w)};
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int row = pass * 16 + (lane >> 2);
                const int qi = pt * 32 + row;
                const int qy = qi / tcw, qx = qi - qy * tcw;
                const int oy = r0 + qy, ox = c0 + qx;
                if (qi < npx && oy < p.H) {
                    float v[8] = {__uint_as_float(lo[pass].x), __uint_as_float(lo[pass].y), __uint_as_float(lo[pass].z),
                                  __uint_as_float(lo[pass].w), __uint_as_float(hi[pass].x), __uint_as_float(hi[pass].y),
                                  __uint_as_float(hi[pass].z), __uint_as_float(hi[pass].w)};
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], scv[e], shv[e]);
                    if (p.act == MV_ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                    } else if (p.act == MV_ACT_GELU_TANH) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
                    }
                    Out8<bf16_t>::st(p.y + (((long long)b * p.H + oy) * p.W + ox) * 64 + half * 32 + c4 * 8, v);
                }
            }
        }
        cur ^= 1;
    }
}

int conv3x3c64_supported(int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw, int in_dtype,
                         int out_dtype, const void* residual, long long M) {
    return C == 64 && K == 64 && R == 3 && S == 3 && sh == 1 && sw == 1 && ph == 1 && pw == 1 && dh == 1 && dw == 1 &&
           in_dtype == MV_BF16 && out_dtype == MV_BF16 && residual == nullptr && M >= 8192;
}

int conv3x3c64_v2_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int H, int W,
                         int act, hipStream_t st) {
    C3V2P p;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.y = (bf16_t*)y;
    p.zero = (const bf16_t*)zero_page(st);
    if (!p.zero) {
        set_error("conv3x3c64_v2: zero page allocation failed");
        return MV_E_OOM;
    }
    p.N = N; p.H = H; p.W = W; p.act = act;
    p.tiles_x = (W + 61) / 62;
    p.tc = (W + p.tiles_x - 1) / p.tiles_x;                 // even column tiles <= 62 (56 -> 56, 112 -> 56, 100 -> 50)
    p.tiles_y = (H + 3) / 4;
    const long long tiles = (long long)N * p.tiles_y * p.tiles_x;
    if (tiles >= (1LL << 31)) {
        set_error("conv3x3c64_v2: too many tiles");
        return MV_E_UNSUPPORTED;
    }
    p.tiles = (int)tiles;
    constexpr int SMEM = 2 * 6 * 64 * 128 + 8 * 32 * (32 * 4 + 16) + 128 * 4;
    const int gx = p.tiles < 256 ? p.tiles : 256;
    set_kernel_name("conv3x3c64_halo");
    auto kern = conv3x3c64_v2_kernel;
    MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    hipLaunchKernelGGL(kern, dim3(gx), dim3(512), SMEM, st, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // namespace mv

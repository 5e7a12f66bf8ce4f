// Two workgroups per CU: the dense-layer GEMM on half the registers and half the LDS of igemm8 (gfx950).
//
// igemm8's 256 x 256 tile needs 128 accumulator registers per wave and 128 KB of LDS, so a CU holds ONE workgroup, and that
// workgroup's prologue (2 us), main loop (18 us at K = 768) and epilogue (5 - 18 us: the CU's memory pipe moves ~25 GB/s,
// about its 1/256 share of HBM) run in series -- the measured "step = matrix time + memory time" of DESIGN.md section 5.2
// (profiles/r04/vit_gemm_tile_phases.txt).  Here a workgroup is sized so that TWO fit on a CU, free-running against each
// other: while one is in its prologue / epilogue (memory pipe, no matrix work) the other's main loop has the matrix pipe.
//
//   * block = 128 rows x 256 channels, 8 waves as two ping-pong groups of four (igemm8s's arrangement): a wave owns 64 x 64
//     (4 MFMA tiles of 32 x 32 = 64 accumulator registers); <= 128 VGPRs, `__launch_bounds__(512, 4)`;
//   * a stage is 32 reduction elements (half of igemm8's k-tile): (128 + 256) rows x 64 bytes = 24 KB, ring of THREE stages =
//     72 KB, `vmcnt(3)`: one whole stage in flight across the barriers.  Per stage and wave: 8 ds_read_b128, 3 LDS-DMA pieces
//     (`buffer_load_dwordx4 ... lds`: 16 rows of 64 bytes per instruction), 8 MFMAs;
//   * LDS rows are 64 bytes, lane-linear for the DMA; the 16-byte chunk a lane fetches is XOR-swizzled on the SOURCE side
//     (chunk ^ ((row >> 2) & 3)) and the fragment reads apply the same XOR: the 16 lanes of a ds_read_b128 group then cover all
//     64 banks once (rows r .. r+3 sit 64 bytes apart: 4 rows x 4 chunk positions);
//   * dense 1 x 1 layers only (Linear, pointwise convolution): no taps, the stage advance is one addition; rows past M and
//     channels past K get an out-of-range DMA offset (the buffer unit writes zeros);
//   * epilogue = igemm8s's (wave-private LDS transpose through the dead ring, full-line stores; scale / shift, residual,
//     ReLU / GELU, fp32 or bf16 out, head-major token output).
// Per FLOP it stages 1.5x the bytes of the 256 x 256 tile; it wins where prologue + epilogue are a large share of a tile
// (short reductions: the ViT / Swin Linears), not on 8192^3.
#include <type_traits>

#include "igemm_pipe.h"

namespace mv {

namespace {

constexpr unsigned OOBH = 0x80000000u;

template <int LOFF>
__device__ __forceinline__ void dma16h(unsigned ldsw, unsigned voff, const u32x4& rsrc, unsigned soff) {
    asm volatile("s_add_u32 m0, %0, %4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(ldsw), "v"(voff), "s"(rsrc), "s"(soff), "n"(LOFF)
                 : "memory", "scc");
}

__device__ __forceinline__ u32x4 make_rsrc_h(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r[2] = OOBH;
    r[3] = 0x00020000u;
    return r;
}

template <int N> __device__ __forceinline__ void wait_vm_lgkm0_h() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

}  // namespace

template <typename OutT, bool LIN>
__global__ __launch_bounds__(512, 4) void igemm8h_kernel(const Igemm2P p) {
    constexpr int BM = 128, BN = 256;
    constexpr int ROWB = 64;                                // bytes per staged row (32 bf16)
    constexpr int XBYTES = BM * ROWB;                       // 8 KB: the x unit comes first in a ring slot, then the w unit
    constexpr int SLOT = (BM + BN) * ROWB;                  // 24 KB
    constexpr int EPITCH = 64 * 4 + 16;
    static_assert(8 * 32 * EPITCH <= 3 * SLOT, "epilogue patches fit the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int xrow0 = 64 * grp, wrow0 = 64 * (wave & 3);
    const int t = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    int tile_m, tile_n;
    tile_coords(t, p.tiles_m, p.tiles_n, tile_m, tile_n, p.gm);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---------------- DMA addressing: one instruction = 64 lanes x 16 bytes = 16 rows of 64 bytes; wave w stages rows 16 w ..
    const int srow = lane >> 2;
    const int gch = (lane & 3) ^ ((srow >> 2) & 3);         // source chunk of LDS slot lane & 3 (swizzle key (row >> 2) & 3)
    const int nk = p.C >> 5;
    const unsigned wrow_bytes = 2u * (unsigned)p.C;
    const u32x4 rx = make_rsrc_h(p.x);
    const u32x4 rw = make_rsrc_h(p.w);
    unsigned xvo, wvo[2];
    {
        const int m = m0 + 16 * wave + srow;
        xvo = m < p.M ? 2u * (unsigned)(m * p.C + gch * 8) : OOBH;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 128 * j + 16 * wave + srow;
            wvo[j] = n < p.K ? (unsigned)n * wrow_bytes + 16u * (unsigned)gch : OOBH;
        }
    }
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + wave * (16 * ROWB));

    auto stage = [&](unsigned soff, auto slotc) {
        constexpr int BASE = decltype(slotc)::value * SLOT;
        dma16h<BASE>(ldsw, xvo, rx, soff);
        dma16h<BASE + XBYTES>(ldsw, wvo[0], rw, soff);
        dma16h<BASE + XBYTES + 8192>(ldsw, wvo[1], rw, soff);
    };

    // ---------------- fragment addressing: slots 1 and 2 are instruction offsets ------------------------------------------
    const int fr = lane & 31, fh = lane >> 5, swz = (fr >> 2) & 3;
    unsigned waddr[2], xaddr[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const unsigned ko = (unsigned)(((2 * kk + fh) ^ swz) << 4);
        xaddr[kk] = lds0 + (xrow0 + fr) * ROWB + ko;
        waddr[kk] = lds0 + XBYTES + (wrow0 + fr) * ROWB + ko;
    }

    const OutT* res = (const OutT*)p.residual;
    ScaleShift8 ss;
    if constexpr (LIN) ss.load_shift(p.shift, n0 + wrow0 + (lane & 7) * 8, p.K);
    else ss.load(p.scale, p.shift, n0 + wrow0 + (lane & 7) * 8, p.K);

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    unsigned soff = 0;                                       // byte offset of the stage being issued inside an operand row
    stage(soff, std::integral_constant<int, 0>{});
    if (nk > 1) {
        soff += 64u;
        stage(soff, std::integral_constant<int, 1>{});
        wait_vm<3>();
    } else {
        wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();              // group 1 runs one barrier behind

    u32x4 wf[2][2], xf[2][2];
    auto kstage = [&](int it, auto slotc) {
        constexpr int SL = decltype(slotc)::value;
        constexpr int OFF = SL * SLOT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            lds_read16<OFF>(wf[0][kk], waddr[kk]);
            lds_read16<OFF + 32 * ROWB>(wf[1][kk], waddr[kk]);
            lds_read16<OFF>(xf[0][kk], xaddr[kk]);
            lds_read16<OFF + 32 * ROWB>(xf[1][kk], xaddr[kk]);
        }
        if (it + 2 < nk) {                                   // stage it+2 goes where stage it-1 was (last read a phase ago)
            soff += 64u;
            stage(soff, std::integral_constant<int, (SL + 2) % 3>{});
            wait_vm_lgkm0_h<3>();                            // stage it+1 has landed
        } else {
            wait_vm_lgkm0_h<0>();
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
            asm volatile("" : "+v"(wf[0][kk]), "+v"(wf[1][kk]), "+v"(xf[0][kk]), "+v"(xf[1][kk]));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[a][kk]),
                                                                        __builtin_bit_cast(bf16x8, xf[b][kk]), acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    for (int it = 0; it < nk; it += 3) {
        kstage(it, std::integral_constant<int, 0>{});
        if (it + 1 < nk) kstage(it + 1, std::integral_constant<int, 1>{});
        if (it + 2 < nk) kstage(it + 2, std::integral_constant<int, 2>{});
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();

    // ---------------- epilogue (igemm8s's; the residual rows of the second pixel tile are requested after the first tile's
    // accumulators have gone to the patch, so that 64 + 32 + 32 registers are live at most) ------------------------------
    char* ep = smem + wave * (32 * EPITCH);
    OutT* y = (OutT*)p.y;
    R8<OutT> late[2][4];
    auto fetch_res = [&](int b, R8<OutT>(&dst)[4]) {
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int m = m0 + xrow0 + b * 32 + pass * 8 + (lane >> 3);
            const int n = n0 + wrow0 + (lane & 7) * 8;
            const bool ok = m < p.M && n < p.K;
            dst[pass].load(res + (ok ? (long long)m * p.K + n : 0));
        }
    };
    if (res) fetch_res(0, late[0]);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = a * 32 + 8 * g + 4 * fh;
                *(float4*)(ep + fr * EPITCH + nl * 4) = make_float4(acc[a][b][4 * g + 0], acc[a][b][4 * g + 1],
                                                                     acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]);
            }
        if (res && b == 0) fetch_res(1, late[1]);
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3), c8 = lane & 7;
            const int m = m0 + xrow0 + b * 32 + row;
            const int n = n0 + wrow0 + c8 * 8;
            const float4 lo = *(const float4*)(ep + row * EPITCH + c8 * 32);
            const float4 hi = *(const float4*)(ep + row * EPITCH + c8 * 32 + 16);
            if (m < p.M && n < p.K) {
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if constexpr (LIN) ss.apply_shift(v);
                else ss.apply(v);
                if (res) late[b][pass].add_to(v);
                if (p.act == MV_ACT_RELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                } else if (p.act == MV_ACT_GELU_TANH && sizeof(OutT) != 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_tanh_f(v[e]);
                }
                long long off = (long long)m * p.K + n;
                if (p.tok > 0) {
                    const int bi = m / p.tok, ti = m - bi * p.tok;
                    off = (((long long)bi * (p.K >> 6) + (n >> 6)) * p.tok + ti) * 64 + (n & 63);
                }
                if constexpr (sizeof(OutT) == 2) {
                    if (p.act == MV_ACT_GELU_TANH) {
                        uint4 u;
                        u.x = gelu_tanh_pack2(v[0], v[1]); u.y = gelu_tanh_pack2(v[2], v[3]);
                        u.z = gelu_tanh_pack2(v[4], v[5]); u.w = gelu_tanh_pack2(v[6], v[7]);
                        *(uint4*)(y + off) = u;
                        continue;
                    }
                }
                Out8<OutT>::st(y + off, v);
            }
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

int igemm8h_supported(long long M, int C, int K, long long x_bytes, long long w_bytes) {
    return C % 64 == 0 && C >= 64 && K % 8 == 0 && M < (1LL << 31) - 256 && x_bytes < (1LL << 31) - (1 << 22) && w_bytes < (1LL << 31);
}

// y[M,K] = act(scale[k] * (x[M,C] . w[K,C]^T) + shift[k] + residual)
int igemm8h_launch(const void* x, const void* w, const float* scale, const float* shift, const void* residual, void* y, long long M,
                   int C, int K, int act, int out_dtype, int tok, hipStream_t st) {
    if (!igemm8h_supported(M, C, K, 2LL * M * C, 2LL * K * C)) {
        set_error("igemm8h: unsupported shape M=%lld C=%d K=%d", M, C, K);
        return MV_E_UNSUPPORTED;
    }
    Igemm2P p;
    memset(&p, 0, sizeof(p));
    p.tok = tok;
    p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.scale = scale; p.shift = shift; p.residual = residual; p.y = y;
    p.N = 1; p.H = (int)M; p.W = 1; p.C = C; p.K = K; p.R = 1; p.S = 1; p.Ho = (int)M; p.Wo = 1;
    p.sh = p.sw = 1; p.dh = p.dw = 1;
    p.M = (int)M;
    p.act = act;
    p.tiles_m = (p.M + 127) / 128;
    p.tiles_n = (K + 255) / 256;
    p.gm = get_flag("i8_gm") ? get_flag("i8_gm") : 8;
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n)), block(512);
    constexpr int SMEM = 3 * 384 * 64;
#define GOH(KERN)                                                                                                   \
    do {                                                                                                            \
        auto kern = KERN;                                                                                           \
        static bool attr = false;                                                                                   \
        if (!attr) {                                                                                                \
            MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));       \
            attr = true;                                                                                            \
        }                                                                                                           \
        hipLaunchKernelGGL(kern, grid, block, SMEM, st, p);                                                         \
    } while (0)
    const bool f32 = out_dtype == MV_F32;
    if (scale) {
        if (f32) GOH((igemm8h_kernel<float, false>));
        else GOH((igemm8h_kernel<bf16_t, false>));
    } else {
        if (f32) GOH((igemm8h_kernel<float, true>));
        else GOH((igemm8h_kernel<bf16_t, true>));
    }
#undef GOH
    set_kernel_name("igemm8h_bf16_128x256_dense");
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // namespace mv

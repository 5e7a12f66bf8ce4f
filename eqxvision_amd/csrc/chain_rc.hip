// ResNet-50 layer 1: the first two bottleneck boundaries WITHOUT the first block's output map in HBM (round 6).
//
// chain1x1.hip fuses the last convolution of bottleneck i with the first one of bottleneck i+1, so the 256-channel block output y_i
// is written once and read once (as block i+1's identity).  Those two passes over y are still 40 % of layer 1's bytes -- and y_0 is
// cheap to RECOMPUTE: it is a pointwise function of two 64-channel maps that block 1's kernel can read instead (reference
// resnet.py:144-162, 295-303):
//
//   y0[m] = relu( [s3 W3_0 | sd Wd] . [t2_0[m] | x0[m]] + shift0 )                  (block 0: conv3 + downsample conv, one GEMM)
//   y1[m] = relu( (s1 W3_1) . t2_1[m] + shift1 + bf16(y0[m]) )                      -> HBM (block 2's identity)
//   t1[m] = relu( (sN W1_2) . bf16(y1[m]) + shiftN )                                -> HBM (block 2's conv2 input)
//
// NPREV = 0 (chain_rc0): block 0's boundary, storing only t1 = relu((sN W1_1) . bf16(y0) + shiftN).
// NPREV = 1 (chain_rc1): block 1's boundary as above: reads t2_1, t2_0, x0 (3 x 64 channels) instead of t2_1 + y0 (64 + 256).
// Together: -205 MB written and -103 MB read per 128 images.  The price is matrix work that these kernels have to spare.
//
// These kernels are not HBM-bound -- block 0's boundary took the same 90 us with and without its 205 MB store -- they are bound by
// what ONE WAVE has to issue per 32-pixel tile: the matrix, vector and LDS instructions of a tile add up (profiles/r06/
// resnet50_layer1_recompute_plan_ab.txt).  So everything here is about a short instruction stream per tile:
//   * weights sit in LDS in MFMA FRAGMENT ORDER (host: ops.chain_rc_fragments): fragment f = 64 lanes x 16 bytes, so the copy in is
//     linear and every A operand is one conflict-free ds_read_b128 at a compile-time offset from an opaque base register;
//   * the whole chain stays in the ACCUMULATOR layout (a lane holds 4 consecutive channels of one pixel per accumulator quad): a
//     bf16 chunk of y IS the B operand of the next conv1 once the host has permuted that layer's reduction index to the
//     accumulator's channel order (k-slot 8 fh + i  <->  channel 8 (2 s + i / 4) + 4 fh + i % 4): no LDS transpose between the GEMMs;
//   * NO per-channel vector arithmetic: every BatchNorm scale is folded into the bf16 weight rows (as chain1x1's DUAL form always
//     did), every shift enters through the matrix pipe as one more k-step -- A = [hi(shift) lo(shift) 0 ...] (two bf16 terms: 16
//     mantissa bits), B = [1 1 0 ...], 4 bytes per lane out of a 256-byte LDS row instead of four 1 KB table reads and 16 adds --
//     and the identity is the C operand that STARTS the next accumulation (y1's chain begins at bf16(y0), not at zero);
//   * ReLU on PACKED bf16 pairs (v_pk_max_i16 against 0: bf16 is sign-magnitude, round(relu(x)) == relu(round(x))): half a
//     conversion + half a max per value is all the vector work a y value costs (this file's first version: 7 instructions);
//   * the weight fragments are fetched SIX matrix instructions ahead into a rolling register ring, and the two 32-channel chunks of
//     a 64-channel slab run INTERLEAVED (independent accumulators sharing every B operand): a wave's consecutive MFMAs wait neither
//     for an LDS round trip nor for each other;
//   * LDS patches are bf16 and only serve the row-major stores (32 rows x 128 bytes): NPREV = 0 stages nothing for y (eight waves
//     per CU), NPREV = 1 runs six (the 128 KB of fragments leave room for six patches; an eight-wave form with 32-channel patches
//     was measured slower).
// Measured alone, 128 images (profiles/r06/resnet50_layer1_recompute_plan_ab.txt): NPREV = 0 41-44 us (chain1x1_dual: 94 with y, 88
// without), NPREV = 1 102-114 us (chain1x1: 109) -- of which ~35 us are its 205 MB of y stores, which do not overlap its 65 us of
// instruction stream (debug-build ablations, same file); resnet50 B = 256: +2.5 ... 3.1 % with the plan on, same box.
#include "mfma_common.h"

namespace mv {

struct ChainRcP {
    const bf16_t* t2;      // NPREV = 1: [M][64] conv2 output of THIS block
    const bf16_t* t2p;     // [M][64] conv2 output of the first block
    const bf16_t* x0;      // [M][64] the stage input (the first block's input)
    const bf16_t* wf;      // 8 * (12 + 4 NPREV) fragments of 1 KB
    const unsigned* sh;    // 8 (1 + NPREV) + 2 shift rows of 64 words: word r < 32 = [hi(shift[row r]) | lo << 16], words 32 .. 63 zero
    bf16_t* y;             // NPREV = 1: [M][256]
    bf16_t* t1;            // [M][64]
    int M, tiles_m;
    int dbg;               // ablations, debug build only ("rc_dbg", results wrong): 1 = y stores dropped, 2 = every tile reads the same 32 rows, 4 = t1 stores dropped
};

// the bf16 pair (relu(a), relu(b)): ONE v_cvt_pk_bf16_f32 + ONE v_pk_max_i16 (bf16 is sign-magnitude: max against 0 as int16 is ReLU).
// The empty asm keeps the packed word a value of its own: without it hipcc turned conversion + vector max into two single conversions
// and a v_perm_b32 per pair.  (Not an asm instruction on purpose: the hazard recogniser does not see into inline asm, and these
// values come straight out of the matrix pipe.)
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t relu_pack_bf2(float a, float b) {
    uint32_t w = pack_bf2(a, b);
    asm volatile("" : "+v"(w));
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, w), z));
}

template <int NPREV, int WAVES, int RD>
__global__ __launch_bounds__(WAVES * 64) void chain_rc_kernel(const ChainRcP p) {
    constexpr int K = 256, N2 = 64, FPC = 12 + 4 * NPREV, NFRAG = 8 * FPC, PITCH = 144, NSH = 8 * (1 + NPREV) + 2, NT = WAVES * 64,
                  NX = 8 + 4 * NPREV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;                                             // NFRAG KB of fragments, then NSH shift rows of 256 bytes
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* ep = smem + NFRAG * 1024 + NSH * 256 + wave * (32 * PITCH);

    {   // fragments + shift rows -> LDS: straight copies, several loads in flight per thread
        constexpr int N16 = NFRAG * 64, U = 4;
        for (int base = 0; base < N16; base += U * NT) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * NT + tid;
                v[u] = ((const uint4*)p.wf)[i < N16 ? i : N16 - 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * NT + tid;
                if (i < N16) ((uint4*)wl)[i] = v[u];
            }
        }
        for (int i = tid; i < NSH * 64; i += NT) ((unsigned*)(wl + NFRAG * 1024))[i] = p.sh[i];
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    typedef __attribute__((address_space(3))) const char* lds_cp;
    // LDS bases as OPAQUE 32-bit addresses: every read below is then base + compile-time immediate.  (Left to itself hipcc
    // materialised one address register per table read and spilled them; scratch reloads count in vmcnt and put
    // `s_waitcnt vmcnt(0)` between the MFMAs.)
    unsigned wbase0 = (unsigned)(uintptr_t)(lds_cp)wl + lane * 16;             // fragments 0 .. 63
    unsigned wbase1 = wbase0 + 65536u;                                            // fragments 64 .. 127 (NPREV = 1)
    unsigned sbase = (unsigned)(uintptr_t)(lds_cp)wl + NFRAG * 1024 + lane * 4;  // shift row r: + 256 r
    asm volatile("" : "+v"(wbase0), "+v"(wbase1), "+v"(sbase));
    auto afrag = [&](int f) -> bf16x8 {
        const lds_cp b = (lds_cp)(uintptr_t)(f < 64 ? wbase0 : wbase1);
        return __builtin_bit_cast(bf16x8, *(const __attribute__((address_space(3))) u32x4_t*)(b + (f & 63) * 1024));
    };
    // the A operand of a shift step: lane r < 32 holds [hi(shift[row r]), lo(shift[row r]), 0 x 6], lanes 32 .. 63 zeros
    auto sfrag = [&](int row) -> bf16x8 {
        u32x4_t v;
        v[0] = *(const __attribute__((address_space(3))) unsigned*)((lds_cp)(uintptr_t)sbase + row * 256);
        v[1] = 0u; v[2] = 0u; v[3] = 0u;
        return __builtin_bit_cast(bf16x8, v);
    };
    u32x4_t onesv;                                               // its B operand: k-slots 0 and 1 (lanes fh = 0) are 1.0
    onesv[0] = fh ? 0u : 0x3f803f80u; onesv[1] = 0u; onesv[2] = 0u; onesv[3] = 0u;
    const bf16x8 ones = __builtin_bit_cast(bf16x8, onesv);

    // xf[0..3] = t2_0, xf[4..7] = x0 (the two K-sources of block 0's GEMM, in Wcat's column order), xf[8..11] = t2_1 (NPREV = 1)
    auto load_x = [&](uint4* xf, int tile) {
        int m = tile * 32 + fr;
        m = m < p.M ? m : p.M - 1;                               // clamp: rows past the end are never stored
        m = (p.dbg & 2) ? fr : m;
        const long long off = (long long)m * 64 + fh * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xf[kk] = *(const uint4*)(p.t2p + off + kk * 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xf[4 + kk] = *(const uint4*)(p.x0 + off + kk * 16);
        if constexpr (NPREV == 1) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xf[8 + kk] = *(const uint4*)(p.t2 + off + kk * 16);
        }
    };

    auto run_tile = [&](uint4* xf, int tile, int refill) {
        const int tile_u = __builtin_amdgcn_readfirstlane(tile);
        const brsrc_t ry = make_brsrc(p.y + (long long)tile_u * 32 * K, NPREV == 1);
        const brsrc_t rt = make_brsrc(p.t1 + (long long)tile_u * 32 * N2);
        const int rows_left = p.M - tile_u * 32;
        f32x16 acc2[2];
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {                         // the next conv1's accumulators start at its shift
            f32x16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.f;
            acc2[a2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sfrag(NSH - 2 + a2), ones, z, 0, 0, 0);
        }
        // The tile's weight fragments are consumed in ONE fixed order (step i -> fragment fid(i)) and fetched RD steps ahead into a
        // rolling register ring; `sched_barrier` pins [take fragment, refill slot, MFMA].  Left to hipcc every fragment was read
        // right before its MFMA: one LDS round trip (~100 cycles) per 32-cycle matrix instruction, ~15k cycles per tile -- the
        // kernel's floor with all memory traffic removed was 77 us (rc_dbg = 7) for 30 us of matrix work.  Per 64-channel slab the
        // two 32-channel chunks' accumulation chains are INTERLEAVED (they are independent and share every B operand), so a wave's
        // consecutive matrix instructions never wait on each other either.
        constexpr int SL = 16 + 8 * NPREV + 8, NSTEP = 4 * SL;
        auto fid = [&](int i) -> int {
            const int ps = i / SL, r = i - ps * SL;
            if (r < 16) return (2 * ps + (r & 1)) * FPC + (r >> 1);
            if (r < 16 + 8 * NPREV) return (2 * ps + ((r - 16) & 1)) * FPC + 8 + ((r - 16) >> 1);
            const int q = r - 16 - 8 * NPREV;
            return (2 * ps + (q >> 2)) * FPC + FPC - 4 + (q & 3);
        };
        bf16x8 ring[RD > 0 ? RD : 1];                            // RD = 0: no ring (hipcc places the reads), for the register-tight form
#pragma unroll
        for (int d = 0; d < RD; ++d) ring[d] = afrag(fid(d));
        auto take = [&](int i) -> bf16x8 {                       // fragment of step i; its slot is refilled for step i + RD
            if constexpr (RD == 0) return afrag(fid(i));
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 a = ring[i % (RD > 0 ? RD : 1)];
            if (i + RD < NSTEP) ring[i % (RD > 0 ? RD : 1)] = afrag(fid(i + RD));
            return a;
        };
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int s0 = ps * SL;
            f32x16 a0[2];
            uint32_t pk[2][8];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
                for (int e = 0; e < 16; ++e) a0[cc][e] = 0.f;
                a0[cc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sfrag(ps * 2 + cc), ones, a0[cc], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bf16x8 a = take(s0 + r);
                a0[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, xf[r >> 1]), a0[r & 1], 0, 0, 0);
                if constexpr (RD > 0) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int g = 0; g < 4; ++g) {                    // y0 = bf16(relu(.)), as block 0 would have stored it
                    pk[cc][2 * g] = relu_pack_bf2(a0[cc][4 * g], a0[cc][4 * g + 1]);
                    pk[cc][2 * g + 1] = relu_pack_bf2(a0[cc][4 * g + 2], a0[cc][4 * g + 3]);
                }
            if constexpr (NPREV == 1) {
                f32x16 a1[2];                                    // y1's accumulation STARTS at y0 (the identity), then shift1, then W3_1 . t2_1
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        a1[cc][4 * g] = __uint_as_float(pk[cc][2 * g] << 16);
                        a1[cc][4 * g + 1] = __uint_as_float(pk[cc][2 * g] & 0xffff0000u);
                        a1[cc][4 * g + 2] = __uint_as_float(pk[cc][2 * g + 1] << 16);
                        a1[cc][4 * g + 3] = __uint_as_float(pk[cc][2 * g + 1] & 0xffff0000u);
                    }
                    a1[cc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sfrag(8 + ps * 2 + cc), ones, a1[cc], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const bf16x8 a = take(s0 + 16 + r);
                    a1[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, xf[8 + (r >> 1)]), a1[r & 1], 0, 0, 0);
                    if constexpr (RD > 0) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        pk[cc][2 * g] = relu_pack_bf2(a1[cc][4 * g], a1[cc][4 * g + 1]);
                        pk[cc][2 * g + 1] = relu_pack_bf2(a1[cc][4 * g + 2], a1[cc][4 * g + 3]);
                        // row-major staging for the store: row = pixel fr, 4 consecutive channels = 8 bytes
                        *(uint2*)(ep + fr * PITCH + (cc * 32 + 8 * g + 4 * fh) * 2) = make_uint2(pk[cc][2 * g], pk[cc][2 * g + 1]);
                    }
            }
            if (ps == 3) load_x(xf, refill);                     // unconditional (rows clamped): a branch here costs the vmcnt count
            // next block's conv1 on this slab: the packed accumulator entries 8 s .. 8 s + 7 of a chunk ARE the B operand of k-step s
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bf16x8 a = take(s0 + 16 + 8 * NPREV + q);
                const int cc = q >> 2, sstep = (q >> 1) & 1;
                const uint4 b = make_uint4(pk[cc][4 * sstep], pk[cc][4 * sstep + 1], pk[cc][4 * sstep + 2], pk[cc][4 * sstep + 3]);
                acc2[q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, b), acc2[q & 1], 0, 0, 0);
                if constexpr (RD > 0) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (NPREV == 1) {
                wave_lds_fence();
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int row = pass * 8 + (lane >> 3);
                    const uint4 u = *(const uint4*)(ep + row * PITCH + (lane & 7) * 16);
                    buf_store_u4(ry, row < rows_left && !(p.dbg & 1) ? (unsigned)(row * K + ps * 64 + (lane & 7) * 8) * 2u : BUF_OOB, u);
                }
                wave_lds_fence();
            }
        }
        // ---- conv1 of the next block: 32 pixels x 64 channels
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(uint2*)(ep + fr * PITCH + (a2 * 32 + 8 * g + 4 * fh) * 2) =
                    make_uint2(relu_pack_bf2(acc2[a2][4 * g], acc2[a2][4 * g + 1]), relu_pack_bf2(acc2[a2][4 * g + 2], acc2[a2][4 * g + 3]));
        wave_lds_fence();
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 8 + (lane >> 3);
            const uint4 u = *(const uint4*)(ep + row * PITCH + (lane & 7) * 16);
            buf_store_u4(rt, row < rows_left && !(p.dbg & 4) ? (unsigned)(row * N2 + (lane & 7) * 8) * 2u : BUF_OOB, u);
        }
        wave_lds_fence();
    };

    uint4 xa[NX], xb[NX];                                       // two pixel tiles in flight per wave
    const int gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
    int tile = gw;
    // The loop body is ALWAYS both tiles (a wave's odd last tile runs behind the loop): with `if (second tile exists)` inside, the
    // two paths into the loop head disagree about what is in flight and hipcc's wait for the first tile's rows becomes
    // vmcnt(11) .. vmcnt(0) -- a drain of the second tile's stores and of its prefetch at the start of every pair.
    load_x(xa, tile);                                           // (rows clamped: harmless for a wave without tiles)
    load_x(xb, tile + nw);
    for (; tile + nw < p.tiles_m; tile += 2 * nw) {
        run_tile(xa, tile, tile + 2 * nw);
        run_tile(xb, tile + nw, tile + 3 * nw);
    }
    if (tile < p.tiles_m) run_tile(xa, tile, tile);
}

int chain_rc_supported(long long M, int C, int K, int N2, int dtype) {
    return dtype == MV_BF16 && C == 64 && K == 256 && N2 == 64 && M >= 8192 && M < (1LL << 31) - (1 << 20) && !get_flag("no_chain") &&
           !get_flag("no_chain_rc");
}

template <int NPREV, int WAVES, int RD>
static int chain_rc_go(ChainRcP& p, hipStream_t st) {
    constexpr int SMEM = 8 * (12 + 4 * NPREV) * 1024 + (8 * (1 + NPREV) + 2) * 256 + WAVES * 32 * 144;
    static_assert(SMEM <= 160 * 1024, "LDS");
    int gx = 256;
    const int need = (p.tiles_m + WAVES - 1) / WAVES;
    if (gx > need) gx = need;
    auto kern = chain_rc_kernel<NPREV, WAVES, RD>;
    static LdsAttrSite attr;
    MV_HIP(attr.ensure((const void*)kern, SMEM));
    hipLaunchKernelGGL(kern, dim3(gx), dim3(WAVES * 64), SMEM, st, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int chain_rc_launch(const void* t2, const void* t2_prev, const void* x0, const void* wfrag, const void* shifts, void* y, void* t1,
                    long long M, hipStream_t st) {
    ChainRcP p;
    p.t2 = (const bf16_t*)t2; p.t2p = (const bf16_t*)t2_prev; p.x0 = (const bf16_t*)x0; p.wf = (const bf16_t*)wfrag;
    p.sh = (const unsigned*)shifts;
    p.y = (bf16_t*)y; p.t1 = (bf16_t*)t1;
    p.M = (int)M;
    p.tiles_m = (int)((M + 31) / 32);
    p.dbg = 0;
#ifdef MV_I8_PROF
    p.dbg = get_flag("rc_dbg");
#endif
    set_kernel_name("chain_rc1_bf16_64x3_256_64");
    return chain_rc_go<1, 6, 6>(p, st);
}

// the first boundary without its output map: t2 = the first block's conv2 output, x0 = the stage input
int chain_rc0_launch(const void* t2, const void* x0, const void* wfrag, const void* shifts, void* t1, long long M, hipStream_t st) {
    ChainRcP p;
    p.t2 = nullptr; p.t2p = (const bf16_t*)t2; p.x0 = (const bf16_t*)x0; p.wf = (const bf16_t*)wfrag; p.sh = (const unsigned*)shifts;
    p.y = nullptr; p.t1 = (bf16_t*)t1;
    p.M = (int)M;
    p.tiles_m = (int)((M + 31) / 32);
    p.dbg = 0;
#ifdef MV_I8_PROF
    p.dbg = get_flag("rc_dbg");
#endif
    set_kernel_name("chain_rc0_bf16_64x2_256_64");
    // eight waves with the fragment ring; twelve (three per SIMD: 168 VGPRs, no room for a ring) measured 45.8 vs 41.2 us
    return chain_rc_go<0, 8, 6>(p, st);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The same style for a boundary whose identity DOES come from memory (the stage's last block: its predecessor's output was written):
//   y [m] = relu( (s3 W3) . t2[m] + shift3 + res[m] )             -> HBM, all pixels or (SUB) only those with even (h, w), compactly
//   t1[m] = relu( (sN W1n) . bf16(y[m]) + shiftN ),  N2 = 128     -> HBM
// = chain1x1_kernel<128, 6, false, SUB> (chain1x1.hip) with this file's instruction stream.  The identity reaches the accumulator
// layout through the wave's LDS patch (row-major 16-byte loads, one slab ahead; 8-byte reads back) and STARTS the accumulation.
struct ChainResP {
    const bf16_t* t2;      // [M][64]
    const bf16_t* res;     // [M][256]
    const bf16_t* wf;      // 8 * (4 + 2 N2 / 32) fragments
    const unsigned* sh;    // 8 + N2 / 32 shift rows
    bf16_t* y;             // [M][256], or SUB: [N][H/2][W/2][256]
    bf16_t* t1;            // [M][N2]
    int M, tiles_m, subH, subW;
};

template <int N2, bool SUB, int WAVES, int RD>
__global__ __launch_bounds__(WAVES * 64) void chain_res_kernel(const ChainResP p) {
    constexpr int K = 256, T2 = N2 / 32, FPC = 4 + 2 * T2, NFRAG = 8 * FPC, PITCH = 144, NSH = 8 + T2, NT = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* ep = smem + NFRAG * 1024 + NSH * 256 + wave * (32 * PITCH);
    {
        constexpr int N16 = NFRAG * 64, U = 4;
        for (int base = 0; base < N16; base += U * NT) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * NT + tid;
                v[u] = ((const uint4*)p.wf)[i < N16 ? i : N16 - 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = base + u * NT + tid;
                if (i < N16) ((uint4*)wl)[i] = v[u];
            }
        }
        for (int i = tid; i < NSH * 64; i += NT) ((unsigned*)(wl + NFRAG * 1024))[i] = p.sh[i];
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    typedef __attribute__((address_space(3))) const char* lds_cp;
    unsigned wbase0 = (unsigned)(uintptr_t)(lds_cp)wl + lane * 16;
    unsigned wbase1 = wbase0 + 65536u;
    unsigned sbase = (unsigned)(uintptr_t)(lds_cp)wl + NFRAG * 1024 + lane * 4;
    asm volatile("" : "+v"(wbase0), "+v"(wbase1), "+v"(sbase));
    auto afrag = [&](int f) -> bf16x8 {
        const lds_cp b = (lds_cp)(uintptr_t)(f < 64 ? wbase0 : wbase1);
        return __builtin_bit_cast(bf16x8, *(const __attribute__((address_space(3))) u32x4_t*)(b + (f & 63) * 1024));
    };
    auto sfrag = [&](int row) -> bf16x8 {
        u32x4_t v;
        v[0] = *(const __attribute__((address_space(3))) unsigned*)((lds_cp)(uintptr_t)sbase + row * 256);
        v[1] = 0u; v[2] = 0u; v[3] = 0u;
        return __builtin_bit_cast(bf16x8, v);
    };
    u32x4_t onesv;
    onesv[0] = fh ? 0u : 0x3f803f80u; onesv[1] = 0u; onesv[2] = 0u; onesv[3] = 0u;
    const bf16x8 ones = __builtin_bit_cast(bf16x8, onesv);

    auto load_x = [&](uint4* xf, int tile) {
        int m = tile * 32 + fr;
        m = m < p.M ? m : p.M - 1;
        const long long off = (long long)m * 64 + fh * 8;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xf[kk] = *(const uint4*)(p.t2 + off + kk * 16);
    };
    // identity rows of 64-channel slab `ps` of a tile, row-major: field i = rows 8 i .. 8 i + 7, 16 bytes per lane.  (A struct of four
    // values handed around BY VALUE: as a `uint4[4]` behind a pointer the rows stayed in scratch memory.)
    struct Rows4 { uint4 r0, r1, r2, r3; };
    auto load_res = [&](int tile, int ps) -> Rows4 {
        auto row = [&](int pass) -> uint4 {
            int m = tile * 32 + pass * 8 + (lane >> 3);
            m = m < p.M ? m : p.M - 1;
            return *(const uint4*)(p.res + (long long)m * K + ps * 64 + (lane & 7) * 8);
        };
        Rows4 o;
        o.r0 = row(0); o.r1 = row(1); o.r2 = row(2); o.r3 = row(3);
        return o;
    };

    // rr: the identity rows of this tile's first slab; returns those of `next_tile`'s
    auto run_tile = [&](uint4* xf, Rows4 rr, int tile, int refill, int next_tile) -> Rows4 {
        const int tile_u = __builtin_amdgcn_readfirstlane(tile);
        const brsrc_t ry = SUB ? make_brsrc(p.y) : make_brsrc(p.y + (long long)tile_u * 32 * K);
        const brsrc_t rt = make_brsrc(p.t1 + (long long)tile_u * 32 * N2);
        const int rows_left = p.M - tile_u * 32;
        unsigned ysub[4] = {0, 0, 0, 0};
        if constexpr (SUB) {
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int m = tile * 32 + pass * 8 + (lane >> 3);
                const int hw = p.subH * p.subW;
                const int b = m / hw, rem = m - b * hw;
                const int h = rem / p.subW, w = rem - h * p.subW;
                const bool keep = m < p.M && !((h | w) & 1);
                ysub[pass] = keep ? (unsigned)(((b * (p.subH >> 1) + (h >> 1)) * (p.subW >> 1) + (w >> 1)) * K) * 2u : BUF_OOB;
            }
        }
        constexpr int SL = 8 + 4 * T2, NSTEP = 4 * SL;
        auto fid = [&](int i) -> int {
            const int ps = i / SL, r = i - ps * SL;
            if (r < 8) return (2 * ps + (r & 1)) * FPC + (r >> 1);
            const int q = r - 8;
            return (2 * ps + q / (2 * T2)) * FPC + 4 + q % (2 * T2);
        };
        bf16x8 ring[RD];
#pragma unroll
        for (int d = 0; d < RD; ++d) ring[d] = afrag(fid(d));
        auto take = [&](int i) -> bf16x8 {
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 a = ring[i % RD];
            if (i + RD < NSTEP) ring[i % RD] = afrag(fid(i + RD));
            return a;
        };
        f32x16 acc2[T2];
#pragma unroll
        for (int a2 = 0; a2 < T2; ++a2) {
            f32x16 z;
#pragma unroll
            for (int e = 0; e < 16; ++e) z[e] = 0.f;
            acc2[a2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sfrag(8 + a2), ones, z, 0, 0, 0);
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int s0 = ps * SL;
            // ---- identity of this slab: row-major registers -> patch -> accumulator layout; the next slab's rows are requested
            {
                char* w = ep + (lane >> 3) * PITCH + (lane & 7) * 16;
                *(uint4*)(w) = rr.r0; *(uint4*)(w + 8 * PITCH) = rr.r1; *(uint4*)(w + 16 * PITCH) = rr.r2; *(uint4*)(w + 24 * PITCH) = rr.r3;
            }
            wave_lds_fence();
            rr = ps < 3 ? load_res(tile, ps + 1) : load_res(next_tile, 0);      // (rows clamped: harmless behind the last tile)
            f32x16 a[2];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const uint2 u = *(const uint2*)(ep + fr * PITCH + (cc * 32 + 8 * g + 4 * fh) * 2);
                    a[cc][4 * g] = __uint_as_float(u.x << 16);
                    a[cc][4 * g + 1] = __uint_as_float(u.x & 0xffff0000u);
                    a[cc][4 * g + 2] = __uint_as_float(u.y << 16);
                    a[cc][4 * g + 3] = __uint_as_float(u.y & 0xffff0000u);
                }
                a[cc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sfrag(ps * 2 + cc), ones, a[cc], 0, 0, 0);
            }
            wave_lds_fence();                                    // the patch is free again (y staging below)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const bf16x8 af = take(s0 + r);
                a[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, xf[r >> 1]), a[r & 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            uint32_t pk[2][8];
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    pk[cc][2 * g] = relu_pack_bf2(a[cc][4 * g], a[cc][4 * g + 1]);
                    pk[cc][2 * g + 1] = relu_pack_bf2(a[cc][4 * g + 2], a[cc][4 * g + 3]);
                    *(uint2*)(ep + fr * PITCH + (cc * 32 + 8 * g + 4 * fh) * 2) = make_uint2(pk[cc][2 * g], pk[cc][2 * g + 1]);
                }
            if (ps == 3) load_x(xf, refill);
#pragma unroll
            for (int q = 0; q < 4 * T2; ++q) {
                const bf16x8 af = take(s0 + 8 + q);
                const int cc = q / (2 * T2), sstep = (q / T2) & 1, a2 = q % T2;
                const uint4 b = make_uint4(pk[cc][4 * sstep], pk[cc][4 * sstep + 1], pk[cc][4 * sstep + 2], pk[cc][4 * sstep + 3]);
                acc2[a2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, b), acc2[a2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            wave_lds_fence();
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = pass * 8 + (lane >> 3);
                const uint4 u = *(const uint4*)(ep + row * PITCH + (lane & 7) * 16);
                if constexpr (SUB) buf_store_u4(ry, ysub[pass] == BUF_OOB ? BUF_OOB : ysub[pass] + (unsigned)(ps * 64 + (lane & 7) * 8) * 2u, u);
                else buf_store_u4(ry, row < rows_left ? (unsigned)(row * K + ps * 64 + (lane & 7) * 8) * 2u : BUF_OOB, u);
            }
            wave_lds_fence();
        }
        // ---- conv1 of the next block: 32 pixels x N2 channels, 64 channels per round through the patch
#pragma unroll
        for (int rd = 0; rd < T2 / 2; ++rd) {
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(uint2*)(ep + fr * PITCH + (a2 * 32 + 8 * g + 4 * fh) * 2) =
                        make_uint2(relu_pack_bf2(acc2[2 * rd + a2][4 * g], acc2[2 * rd + a2][4 * g + 1]),
                                   relu_pack_bf2(acc2[2 * rd + a2][4 * g + 2], acc2[2 * rd + a2][4 * g + 3]));
            wave_lds_fence();
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int row = pass * 8 + (lane >> 3);
                const uint4 u = *(const uint4*)(ep + row * PITCH + (lane & 7) * 16);
                buf_store_u4(rt, row < rows_left ? (unsigned)(row * N2 + rd * 64 + (lane & 7) * 8) * 2u : BUF_OOB, u);
            }
            wave_lds_fence();
        }
        return rr;
    };

    uint4 xa[4], xb[4];
    const int gw = blockIdx.x * WAVES + wave, nw = gridDim.x * WAVES;
    int tile = gw;
    load_x(xa, tile);
    load_x(xb, tile + nw);
    Rows4 rr = load_res(tile, 0);
    for (; tile + nw < p.tiles_m; tile += 2 * nw) {
        rr = run_tile(xa, rr, tile, tile + 2 * nw, tile + nw);
        rr = run_tile(xb, rr, tile + nw, tile + 3 * nw, tile + 2 * nw);
    }
    if (tile < p.tiles_m) run_tile(xa, rr, tile, tile, tile);
}

int chain_res_supported(long long N, int H, int W, int C, int K, int N2, int sub, int dtype) {
    const long long M = N * H * W;
    if (!(dtype == MV_BF16 && C == 64 && K == 256 && N2 == 128 && M >= 8192 && M < (1LL << 31) - (1 << 20)) || get_flag("no_chain") ||
        get_flag("no_chain_res"))
        return 0;
    if (sub == 0) return 1;
    return sub == 2 && H % 2 == 0 && W % 2 == 0 && N * (H / 2) * (W / 2) * K * 2 < (1LL << 31) && !get_flag("no_chain_sub");
}

int chain_res_launch(const void* t2, const void* residual, const void* wfrag, const void* shifts, void* y, void* t1, int N, int H, int W,
                     int sub, hipStream_t st) {
    constexpr int WAVES = 8, N2 = 128, RD = 6;
    constexpr int SMEM = 8 * (4 + 2 * N2 / 32) * 1024 + (8 + N2 / 32) * 256 + WAVES * 32 * 144;
    static_assert(SMEM <= 160 * 1024, "LDS");
    ChainResP p;
    p.t2 = (const bf16_t*)t2; p.res = (const bf16_t*)residual; p.wf = (const bf16_t*)wfrag; p.sh = (const unsigned*)shifts;
    p.y = (bf16_t*)y; p.t1 = (bf16_t*)t1;
    const long long M = (long long)N * H * W;
    p.M = (int)M;
    p.tiles_m = (int)((M + 31) / 32);
    p.subH = H; p.subW = W;
    int gx = 256;
    const int need = (p.tiles_m + WAVES - 1) / WAVES;
    if (gx > need) gx = need;
    if (sub) {
        auto kern = chain_res_kernel<N2, true, WAVES, RD>;
        static LdsAttrSite attr;
        MV_HIP(attr.ensure((const void*)kern, SMEM));
        set_kernel_name("chain_res_bf16_64_256_128_ysub2");
        hipLaunchKernelGGL(kern, dim3(gx), dim3(WAVES * 64), SMEM, st, p);
    } else {
        auto kern = chain_res_kernel<N2, false, WAVES, RD>;
        static LdsAttrSite attr;
        MV_HIP(attr.ensure((const void*)kern, SMEM));
        set_kernel_name("chain_res_bf16_64_256_128");
        hipLaunchKernelGGL(kern, dim3(gx), dim3(WAVES * 64), SMEM, st, p);
    }
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // namespace mv

// ResNet bottleneck tail, CU-resident per image (gfx950):
//     y = relu(bn3(conv3_1x1(relu(bn2(conv2_3x3(t1))))) + identity)            (reference resnet.py:144-162)
// in ONE launch.  One 512-thread block owns one image (14 x 14 maps: 256 images <-> 256 CUs, one round, no tile
// quantisation): the 3x3 convolution reads its whole input map from LDS, its output map goes back to LDS as bf16 and
// is the operand of the 1x1 expansion -- the `width`-channel intermediate never reaches HBM, and the halo of the 3x3
// costs nothing because the whole image is resident.
//
//   * positions are enumerated on a padded grid of pitch 16: o = 16 * oy + ox (ox = 14, 15 are dead columns), so the
//     14 x 14 map is exactly 7 MFMA column blocks of 32 positions, and the input of filter tap (dy, dx) for position o
//     is LDS slot o + 17 + 16 dy + dx of the zero-bordered map (slot = 16 (y + 1) + (x + 1)): ONE constant offset per
//     tap, no bounds checks in the main loop;
//   * a wave owns 32 output channels x all 224 positions (7 accumulator tiles = 112 registers): A = weights (rows =
//     channels) come STRAIGHT from L2 into registers -- they are wave-private, pre-arranged by the host in fragment
//     order so that every load is one coalesced 1 KB piece (rolling prefetch, 4 / 8 k-steps ahead); B = positions come
//     from LDS (shared by the 8 waves), rows of `width` channels with the 16-byte chunk index XOR-swizzled by
//     (slot & 15): conflict-free ds_read_b128;
//   * no barrier inside the main loops: the maps are static in LDS, the waves free-run (two per SIMD);
//   * the 1x1 expansion runs per 256-channel chunk in two position halves (64 / 48 accumulator registers) so that the
//     residual rows of the half (fetched row-major at its start, behind the 8 weight fragments already in flight)
//     fit in registers; epilogue = wave-private LDS transpose, fp32 scale / shift + residual + ReLU, 16-byte stores.
#include "mfma_common.h"

namespace mv {

namespace {

struct BneckP {
    const bf16_t* t1;     // [B][HW][HW][WID] NHWC
    const bf16_t* w2f;    // fragment order [8 waves][9 taps][WID/16][64 lanes][8]
    const float* s2;      // [WID]
    const float* h2;
    const bf16_t* w3f;    // fragment order [COUT/256][8 waves][WID/16][64 lanes][8]
    const float* s3;      // [COUT]
    const float* h3;
    const bf16_t* res;    // [B][HW][HW][COUT]
    bf16_t* y;
    int skew;             // conv3: waves 4-7 start `skew` x 512 cycles late (0 = in phase)
    long long* prof;      // experiments only (tools/time_bneck.py --prof): per-wave wall-clock / shader-clock stamps at the phase boundaries
};

template <int N> struct IC { static constexpr int value = N; };

template <int WID, int COUT, int HW>
__global__ __launch_bounds__(512) void bneck_tail_kernel(const BneckP p) {
    constexpr int PITCH = 16;
    static_assert(HW <= 14 && (HW * PITCH) % 32 == 0, "padded map must be whole 32-position blocks");
    static_assert(WID == 256, "a wave owns 32 of 256 channels");
    constexpr int NB = HW * PITCH / 32;            // 7 position blocks
    constexpr int ROWB = WID * 2;                  // bytes per position row
    constexpr int CPR = ROWB / 16;                 // 16-byte chunks per row
    constexpr int KS = WID / 16;                   // k16-steps per filter tap
    constexpr int NSLOT = NB * 32 + 2 * PITCH + 2;
    constexpr int T2B = NB * 32 * ROWB;
    constexpr int EPITCH = 32 * 4 + 16;
    constexpr int NCHUNK = COUT / 256;
    constexpr int NPIX = HW * HW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int img = blockIdx.x;
    const int fr = lane & 31, fh = lane >> 5;
    long long st_w[6];
    unsigned st_c[6];
#define MV_BT_STAMP(i)                                                   \
    do {                                                                 \
        if (p.prof) {                                                    \
            st_w[i] = wall_clock64();                                    \
            st_c[i] = (unsigned)__builtin_readcyclecounter();            \
        }                                                                \
    } while (0)
    MV_BT_STAMP(0);

    // ---------------- the first weight fragments of both products (oldest loads) --------------------------------
    constexpr int D2 = 4, D3 = 8;
    constexpr int NSTEP2 = 9 * KS;
    const uint4* ap2 = (const uint4*)p.w2f + (size_t)wave * NSTEP2 * 64 + lane;
    uint4 a2[D2];
#pragma unroll
    for (int d = 0; d < D2; ++d) a2[d] = ap2[d * 64];

    // ---------------- phase 0: t1 -> LDS (zero border, swizzled rows) ------------------------------------------
    {
        constexpr int NCH = NPIX * CPR;
        constexpr int NIT = (NCH + 511) / 512;
        const uint4* tg = (const uint4*)(p.t1 + (size_t)img * NPIX * WID);
        uint4 v[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + 512 * i;
            v[i] = tg[idx < NCH ? idx : 0];
        }
        for (int idx = tid; idx < NSLOT * CPR; idx += 512) {
            const int s = idx / CPR;
            const int Y = s >> 4, X = s & 15;
            if (!(Y >= 1 && Y <= HW && X >= 1 && X <= HW)) *(uint4*)(smem + idx * 16) = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + 512 * i;
            if (idx < NCH) {
                const int pos = idx / CPR, c = idx - pos * CPR;
                const int yy = pos / HW, xx = pos - yy * HW;
                const int s = (yy + 1) * PITCH + xx + 1;
                *(uint4*)(smem + s * ROWB + ((c ^ (s & 15)) << 4)) = v[i];
            }
        }
    }
    __syncthreads();
    MV_BT_STAMP(1);

    // ---------------- conv2: 3x3 over the resident map -----------------------------------------------------------
    f32x16 acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;

    const int lb = (fr + PITCH + 1) * ROWB;
    auto bptr2 = [&](int tap, int j) -> const char* {
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int sw = (fr + 1 + dx) & 15;
        return smem + lb + (dy * PITCH + dx) * ROWB + (((2 * j + fh) ^ sw) << 4);
    };
    {
        bf16x8 bc[NB], bn[NB];
        {
            const char* q = bptr2(0, 0);
#pragma unroll
            for (int b = 0; b < NB; ++b) bc[b] = *(const bf16x8*)(q + b * 32 * ROWB);
        }
        for (int tap = 0; tap < 9; ++tap) {
            const int tapn = tap < 8 ? tap + 1 : 8;
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                // one k16-step: the weight fragment D2 steps ahead is requested, then the 7 MFMAs of this step, each followed by the
                // read of the NEXT step's fragment of the same position block (pinned pairs: left alone, hipcc pairs every ds_read
                // with a wait + its MFMA and batches the weight loads four at a time; all 7 reads up front make the 8 waves hit
                // the LDS in bursts while the matrix pipe idles)
                const char* q = (j == KS - 1) ? bptr2(tapn, 0) : bptr2(tap, j + 1);
                int nxt = tap * KS + j + D2;
                nxt = nxt < NSTEP2 ? nxt : NSTEP2 - 1;
                const uint4* an = ap2 + (size_t)nxt * 64;
                __builtin_amdgcn_sched_barrier(0);
                const bf16x8 af = __builtin_bit_cast(bf16x8, a2[j % D2]);
                a2[j % D2] = *an;
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bc[b], acc[b], 0, 0, 0);
                    bn[b] = *(const bf16x8*)(q + b * 32 * ROWB);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int b = 0; b < NB; ++b) bc[b] = bn[b];
            }
        }
    }

    MV_BT_STAMP(2);
    // ---------------- first fragments of the expansion (they land during the epilogue below) ---------------------
    constexpr int NSTEP3 = NCHUNK * 2 * KS;                  // linear step counter over (chunk, half, k-step)
    const uint4* ap3 = (const uint4*)p.w3f + lane;
    auto aidx3 = [&](int t) -> size_t {
        t = t < NSTEP3 ? t : NSTEP3 - 1;
        const int cn = t / (2 * KS), j = t % KS;
        return ((size_t)(cn * 8 + wave) * KS + j) * 64;
    };
    uint4 a3[D3];
#pragma unroll
    for (int d = 0; d < D3; ++d) a3[d] = ap3[aidx3(d)];

    // ---------------- conv2 epilogue: BN + ReLU -> bf16 -> t2 in LDS (over t1) -----------------------------------
    {
        float4 sc2[4], sh2[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = 32 * wave + 8 * g + 4 * fh;
            sc2[g] = *(const float4*)(p.s2 + ch);
            sh2[g] = *(const float4*)(p.h2 + ch);
        }
        __syncthreads();                                     // every wave has finished reading t1
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float v0 = fmaxf(fmaf(acc[b][4 * g + 0], sc2[g].x, sh2[g].x), 0.f);
                const float v1 = fmaxf(fmaf(acc[b][4 * g + 1], sc2[g].y, sh2[g].y), 0.f);
                const float v2 = fmaxf(fmaf(acc[b][4 * g + 2], sc2[g].z, sh2[g].z), 0.f);
                const float v3 = fmaxf(fmaf(acc[b][4 * g + 3], sc2[g].w, sh2[g].w), 0.f);
                uint2 u;
                u.x = pack_bf2(v0, v1);
                u.y = pack_bf2(v2, v3);
                const int o = 32 * b + fr;
                *(uint2*)(smem + o * ROWB + (((4 * wave + g) ^ (fr & 15)) << 4) + 8 * fh) = u;
            }
        __syncthreads();
    }
    MV_BT_STAMP(3);

    // ---------------- conv3: 1x1 expansion, per 256-channel chunk and position half ------------------------------
    // Order of a wave's memory requests (they complete in order): the weight fragments run D3 = 8 steps ahead of their use,
    // straight through the half boundaries; the residual rows of half h+1 are requested DURING the epilogue of half h, block
    // by block into the registers that block's epilogue has just released -- so they are younger than the 8 fragments the next
    // main loop starts on and have an epilogue plus 8 k-steps to arrive before a fragment queued behind them is needed.
    char* ep = smem + T2B + wave * (32 * EPITCH);
    const int er = lane >> 2, ec = lane & 3;
    const bf16_t* resi = p.res + (size_t)img * NPIX * COUT;
    bf16_t* yi = p.y + (size_t)img * NPIX * COUT;
    const int lb3 = fr * ROWB;
    const int sw3 = fr & 15;
    constexpr int NBA = (NB + 1) / 2, NBB = NB / 2;          // blocks of the two halves: 4 + 3

    uint4 rr[NBA][2];                                        // residual rows of the half whose epilogue comes next
    auto pix_of = [&](int blk, int ps) -> int {              // image pixel of this lane's row in pass ps of position block blk
        int o = 32 * blk + 16 * ps + er;
        asm volatile("" : "+v"(o));                          // recomputed where used: hoisted, the 14 row addresses spill
        const int oy = o >> 4, ox = o & 15;
        return ox < HW ? oy * HW + ox : -1;
    };
    auto fetch_res = [&](int slot, int blk, int cn) {
        const int n = cn * 256 + wave * 32 + ec * 8;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            const int pix = pix_of(blk, ps);
            rr[slot][ps] = *(const uint4*)(resi + (size_t)(pix < 0 ? 0 : pix) * COUT + n);
        }
    };
#pragma unroll
    for (int b = 0; b < NBA; ++b) fetch_res(b, b, 0);        // half (chunk 0, A)
    if (p.skew > 0 && wave >= 4)                             // the second wave of every SIMD starts half a period late: one in
        for (int i = 0; i < p.skew; ++i) __builtin_amdgcn_s_sleep(8);     // its main loop while the other is in its epilogue

    auto do_half = [&](auto b0c, auto nbc, auto b0n, auto nbn, int cn, int cn_next, int t0, const float4& sca, const float4& scb,
                       const float4& sha, const float4& shb) {
        constexpr int B0 = decltype(b0c)::value, NBH = decltype(nbc)::value;
        constexpr int B0N = decltype(b0n)::value, NBN = decltype(nbn)::value;      // the half that follows (cn_next < 0: none)
        const int n = cn * 256 + wave * 32 + ec * 8;
        f32x16 c3[NBH];
#pragma unroll
        for (int b = 0; b < NBH; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) c3[b][e] = 0.f;
        bf16x8 bc[NBH], bn[NBH];
        const char* q0 = smem + lb3 + B0 * 32 * ROWB;
#pragma unroll
        for (int b = 0; b < NBH; ++b) bc[b] = *(const bf16x8*)(q0 + ((fh ^ sw3) << 4) + b * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int jn = j + 1 < KS ? j + 1 : j;
            const uint4* an = ap3 + aidx3(t0 + j + D3);
            const char* qn = q0 + (((2 * jn + fh) ^ sw3) << 4);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8 af = __builtin_bit_cast(bf16x8, a3[j % D3]);
            a3[j % D3] = *an;
#pragma unroll
            for (int b = 0; b < NBH; ++b) {
                c3[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bc[b], c3[b], 0, 0, 0);
                bn[b] = *(const bf16x8*)(qn + b * 32 * ROWB);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int b = 0; b < NBH; ++b) bc[b] = bn[b];
        }
        // residual slots the next half uses and this epilogue does not: requested right away
        if (cn_next >= 0) {
#pragma unroll
            for (int b = NBH; b < NBN; ++b) fetch_res(b, B0N + b, cn_next);
        }
        // epilogue: transpose through the wave's patch, BN + identity + ReLU, 16-byte stores
#pragma unroll
        for (int b = 0; b < NBH; ++b) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(ep + fr * EPITCH + (8 * g + 4 * fh) * 4) =
                    make_float4(c3[b][4 * g + 0], c3[b][4 * g + 1], c3[b][4 * g + 2], c3[b][4 * g + 3]);
            wave_lds_fence();
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int row = 16 * ps + er;
                const float4 lo = *(const float4*)(ep + row * EPITCH + ec * 32);
                const float4 hi = *(const float4*)(ep + row * EPITCH + ec * 32 + 16);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                v[0] = fmaf(v[0], sca.x, sha.x); v[1] = fmaf(v[1], sca.y, sha.y);
                v[2] = fmaf(v[2], sca.z, sha.z); v[3] = fmaf(v[3], sca.w, sha.w);
                v[4] = fmaf(v[4], scb.x, shb.x); v[5] = fmaf(v[5], scb.y, shb.y);
                v[6] = fmaf(v[6], scb.z, shb.z); v[7] = fmaf(v[7], scb.w, shb.w);
                const uint32_t w[4] = {rr[b][ps].x, rr[b][ps].y, rr[b][ps].z, rr[b][ps].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += __uint_as_float(w[e] << 16);
                    v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                const int pix = pix_of(B0 + b, ps);
                if (pix >= 0) Out8<bf16_t>::st(yi + (size_t)pix * COUT + n, v);
            }
            wave_lds_fence();
            if (cn_next >= 0 && b < NBN) fetch_res(b, B0N + b, cn_next);     // this block's registers are free again
        }
    };

    for (int cn = 0; cn < NCHUNK; ++cn) {
        const int n = cn * 256 + wave * 32 + ec * 8;
        const float4 sca = *(const float4*)(p.s3 + n), scb = *(const float4*)(p.s3 + n + 4);
        const float4 sha = *(const float4*)(p.h3 + n), shb = *(const float4*)(p.h3 + n + 4);
        do_half(IC<0>{}, IC<NBA>{}, IC<NBA>{}, IC<NBB>{}, cn, cn, (2 * cn) * KS, sca, scb, sha, shb);
        do_half(IC<NBA>{}, IC<NBB>{}, IC<0>{}, IC<NBA>{}, cn, cn + 1 < NCHUNK ? cn + 1 : -1, (2 * cn + 1) * KS, sca, scb, sha, shb);
        if (cn == 0) MV_BT_STAMP(4);
    }
    MV_BT_STAMP(5);
    if (p.prof && lane == 0) {
        long long* o = p.prof + ((size_t)blockIdx.x * 8 + wave) * 12;
#pragma unroll
        for (int i = 0; i < 6; ++i) { o[i] = st_w[i]; o[6 + i] = (long long)st_c[i]; }
    }
#undef MV_BT_STAMP
}

}  // namespace

}  // namespace mv

extern "C" {

int mv_bottleneck_tail_supported(int H, int W, int width, int cout, int dtype) {
    if (mv::get_flag("no_bneck_tail")) return 0;
    return dtype == MV_BF16 && H == 14 && W == 14 && width == 256 && cout == 1024;
}

int mv_bottleneck_tail_fwd(const void* t1, const void* w2f, const float* scale2, const float* shift2, const void* w3f,
                           const float* scale3, const float* shift3, const void* residual, void* y, int B, int H, int W,
                           int width, int cout, int dtype, mv_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    using namespace mv;
    MV_CHECK_ARG(t1 && w2f && scale2 && shift2 && w3f && scale3 && shift3 && residual && y,
                 "mv_bottleneck_tail_fwd: null argument");
    MV_CHECK_ARG(B > 0, "mv_bottleneck_tail_fwd: B = %d", B);
    if (!mv_bottleneck_tail_supported(H, W, width, cout, dtype)) {
        set_error("mv_bottleneck_tail_fwd: unsupported configuration %dx%d width %d -> %d (ask mv_bottleneck_tail_supported first)",
                  H, W, width, cout);
        return MV_E_UNSUPPORTED;
    }
    BneckP p;
    p.t1 = (const bf16_t*)t1; p.w2f = (const bf16_t*)w2f; p.s2 = scale2; p.h2 = shift2;
    p.w3f = (const bf16_t*)w3f; p.s3 = scale3; p.h3 = shift3; p.res = (const bf16_t*)residual; p.y = (bf16_t*)y;
    p.skew = 0;                 // the skewed two-half schedule measured +0 / -2 %: the switch is gone
    p.prof = nullptr;
#ifdef MV_I8_PROF              // debug build only (EQV_PROF=1): the production library never turns a flag into a device address
    if (get_flag("bneck_prof"))
        p.prof = (long long*)(((unsigned long long)(unsigned)get_flag("prof_hi") << 32) | (unsigned)get_flag("prof_lo"));
#endif
    constexpr int SMEM = 7 * 32 * 512 + 8 * 32 * 144;        // t2 + the 8 epilogue patches (>= the zero-bordered t1 map)
    static_assert(SMEM >= (7 * 32 + 34) * 512, "t1 map must fit");
    auto kern = bneck_tail_kernel<256, 1024, 14>;
    static LdsAttrSite attr;
    MV_HIP(attr.ensure((const void*)kern, SMEM));
    set_kernel_name("bneck_tail_bf16_14x14_256_1024");
    hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(512), SMEM, stream, p);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

// Fused multi-head attention core for gfx950 (reference _VitAttention.__call__, vit.py:65-73):
//   out[b, i, h*dh + d] = sum_j softmax_j( scale * q_i . k_j ) v_j[d]
// reading q/k/v straight out of the qkv Linear output [B, N, 3, H, dh] (bf16).
//
// One workgroup (4 waves) per (image, head).  K (row-major, padded pitch) and V^T (key-contiguous,
// padded pitch) of the head live in LDS; each wave walks 32-query tiles:
//   S^T = K . Q^T      v_mfma_f32_32x32x16_bf16, A = K rows (LDS), B = Q rows (global -> VGPR)
//   softmax            the accumulator layout puts ONE query per lane (column) and its keys across
//                      the lane's registers + the partner lane (lane^32): max / sum are register
//                      reductions plus a single cross-half exchange, no LDS, no serial lanes
//   O^T = V^T . P^T    the MFMA reduction index only has to agree between A and B, so key order
//                      inside a k16-step is CHOSEN to be the accumulator's own register order: the
//                      bf16-packed probabilities feed the B operand with no cross-lane movement
//                      and V^T fragments are two 8-byte LDS reads per lane.
// Whole-row softmax (N <= 256 keys held in accumulators): no online rescaling needed for ViT's
// N = 197.  Longer sequences / odd head sizes use the generic kernel.
#include "common.h"
#include "rng_common.h"

namespace mv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DH, int NT, bool HM, bool PR>
__global__ __launch_bounds__(512) void mha_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                      float* __restrict__ probs, int B, int N, int H, float scale,
                                                      const uint32_t* __restrict__ drop_keys, float keep) {
    constexpr int NP = NT * 32;                                     // padded key count
    constexpr int KPITCH = DH * 2 + 16;                             // bytes; odd number of 16-B slots
    constexpr int VPITCH = NP * 2 + 16;                             // bytes; 16 * odd: 16-byte aligned, conflict-free b128 rows
    constexpr int KC = DH / 16;                                     // k16 steps of Q.K^T
    constexpr int DT = DH / 32;                                     // 32-wide d tiles of the output
    __shared__ __attribute__((aligned(16))) char smem[NP * KPITCH + DH * VPITCH];
    char* kl = smem;
    char* vl = smem + NP * KPITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int nqt = (N + 31) / 32;
    const float sl2 = scale * 1.4426950408889634f;   // exp(scale*x) = exp2(sl2*x)
    // token-major qkv [B, N, 3, H, dh] (the Linear's own output: 128-byte pieces at a 3*H*dh row stride) or
    // head-major [B, 3, H, N, dh] (every head contiguous; written by mv_linear_heads_fwd)
    const long long rs = HM ? DH : 3LL * H * DH;                     // row stride (elements)
    const long long hs = HM ? (long long)N * DH : DH;                // head stride
    const int npairs = B * H;
    auto q_of = [&](int pair) {
        const int b = pair / H, h = pair - b * H;
        return qkv + (long long)b * N * 3 * H * DH + (long long)h * hs;
    };

    // One 8-wave block per CU, PERSISTENT over (image, head) pairs; each wave owns one 32-query tile of the
    // pair (ViT: 7 tiles).  K / V of the NEXT pair are fetched into registers right after the current pair's
    // copy has been written to LDS (4 + 4 x 16 bytes per thread), and the Q fragments run one tile ahead across
    // pair boundaries, so no global latency is exposed.  The one-shot version (a block per pair, two blocks
    // per CU starting -- and stalling -- in lockstep) left the SIMDs idle ~55% of the time
    // (PMC: SQ_ACTIVE_INST_ANY vs SQ_WAVE_CYCLES).
    constexpr int CH = DH / 8;
    constexpr int KITEMS = NP * CH;
    constexpr int NTHR = 512, NW = NTHR / 64;
    constexpr int KI = (KITEMS + NTHR - 1) / NTHR;
    constexpr int VITEMS = (NP / 2) * CH;
    constexpr int VI = (VITEMS + NTHR - 1) / NTHR;
    uint4 kv[KI], v0[VI], v1[VI];
    int stid = tid;                        // staging index; re-laundered every pair (see the loop) so that the
                                           // per-item addresses are recomputed instead of living in VGPRs
    auto load_kv = [&](int pair) {        // unconditional loads from clamped rows (zeroed when written to LDS)
        const bf16_t* kbase = q_of(pair) + (long long)H * hs;
        const bf16_t* vbase = kbase + (long long)H * hs;
#pragma unroll
        for (int j = 0; j < KI; ++j) {
            const int i = j * NTHR + stid;
            const int key = i / CH, ch = i - key * CH;
            const int kc = key < N ? key : N - 1;
            kv[j] = *(const uint4*)(kbase + (long long)kc * rs + ch * 8);
        }
#pragma unroll
        for (int j = 0; j < VI; ++j) {
            const int i = j * NTHR + stid;
            const int ch = i / (NP / 2), kp = i - ch * (NP / 2);      // consecutive lanes -> consecutive key pairs
            const int key = 2 * kp;
            const int k0 = key < N ? key : N - 1, k1 = key + 1 < N ? key + 1 : N - 1;
            const int chc = ch < CH ? ch : CH - 1;
            v0[j] = *(const uint4*)(vbase + (long long)k0 * rs + chc * 8);
            v1[j] = *(const uint4*)(vbase + (long long)k1 * rs + chc * 8);
        }
    };
    auto load_q = [&](uint4* qf, int pair, int qt_) {
        int qq = qt_ * 32 + fr;
        qq = qq < N ? qq : N - 1;
        const bf16_t* qp = q_of(pair) + (long long)qq * rs + fh * 8;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) qf[kk] = *(const uint4*)(qp + kk * 16);
    };
    uint4 qcur[KC];
    int pair = blockIdx.x;
    if (pair >= npairs) return;
    load_kv(pair);
    load_q(qcur, pair, wave < nqt ? wave : 0);

    for (; pair < npairs; pair += gridDim.x) {
        const int b = pair / H, h = pair - b * H;
        const int npair = pair + (int)gridDim.x < npairs ? pair + (int)gridDim.x : pair;   // clamped: harmless re-read
        asm volatile("" : "+v"(stid));
        // ---- registers -> LDS: K [key][d], V^T [d][key] (2-key x 8-d patches transposed into dword stores)
        {
#pragma unroll
        for (int j = 0; j < KI; ++j) {
            const int i = j * NTHR + stid;
            const int key = i / CH, ch = i - key * CH;
            if (i < KITEMS) *(uint4*)(kl + key * KPITCH + ch * 16) = key < N ? kv[j] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < VI; ++j) {
            const int i = j * NTHR + stid;
            if (i >= VITEMS) continue;
            const int ch = i / (NP / 2), kp = i - ch * (NP / 2);
            const int key = 2 * kp;
            // keys of a 16-group are stored in the order the P.V MFMA consumes them (bits 2 and 3 of the key
            // swapped: 0-3, 8-11 | 4-7, 12-15), so a lane's 8 keys are ONE 16-byte LDS read
            const int kpos = (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1);
            const uint4 z = make_uint4(0, 0, 0, 0);
            const uint4 w0 = key < N ? v0[j] : z, w1 = key + 1 < N ? v1[j] : z;
            const uint32_t a[4] = {w0.x, w0.y, w0.z, w0.w}, c[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t lo = (a[e] & 0xffffu) | (c[e] << 16);          // d = 8ch+2e   : (key, key+1)
                const uint32_t hi = (a[e] >> 16) | (c[e] & 0xffff0000u);      // d = 8ch+2e+1
                *(uint32_t*)(vl + (ch * 8 + 2 * e) * VPITCH + kpos * 2) = lo;
                *(uint32_t*)(vl + (ch * 8 + 2 * e + 1) * VPITCH + kpos * 2) = hi;
            }
        }
        }
        __syncthreads();
        load_kv(npair);                          // in flight during this pair's tiles

        // NT <= 8 = NW: a wave has at most ONE tile per pair.  (Written as a loop, the never-taken back-edge made
        // hipcc wait for the K/V prefetch just issued -- s_waitcnt vmcnt(3) -- before the first Q.K^T MFMA.)
        static_assert(NT <= NW, "one query tile per wave");
        if (wave < nqt) {
            const int qt = wave;
            const int q = qt * 32 + fr;
            const bool qvalid = q < N;
            // ---- S^T = K . Q^T: 28 (k16-step, key tile) MFMAs, the K fragment of step i + SD is read from
            //      LDS before MFMA i issues (pinned with sched barriers: left alone, hipcc either waits for every
            //      read right where it is issued or hoists all of them and spills) ------------------------------
            f32x16 sacc[NT];
            {
                constexpr int NS = KC * NT, SD = 4;
                uint4 kf[NS];
                auto rdk = [&](int i) {
                    const int kk = i / NT, kt = i - kk * NT;
                    kf[i] = *(const uint4*)(kl + (kt * 32 + fr) * KPITCH + kk * 32 + fh * 16);
                };
#pragma unroll
                for (int i = 0; i < SD && i < NS; ++i) rdk(i);
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int kk = i / NT, kt = i - kk * NT;
                    if (i + SD < NS) rdk(i + SD);
                    if (kk == 0) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) sacc[kt][e] = 0.f;
                    }
                    sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[i]),
                                                                       __builtin_bit_cast(bf16x8, qcur[kk]), sacc[kt], 0,
                                                                       0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- softmax over keys: lane = query column; keys = (kt, reg) and the partner half ------
            // accumulator register e of tile kt holds key 32*kt + (e&3) + 8*(e>>2) + 4*fh
            // (measured: splitting the max / sum chains four ways, or packed v_pk_fma/v_pk_add, is SLOWER here)
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = sacc[kt][e];
                    if (kt == NT - 1) {   // only the last key tile can hold padded keys (NP - N < 32)
                        const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                        v = key < N ? v : -INFINITY;
                        sacc[kt][e] = v;
                    }
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float nmx = -mx * sl2;
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    // one fma + the raw v_exp_f32 (exp2f() wraps it in ~5 denormal-range fix-up instructions and
                    // this loop is the VALU bulk of the kernel); masked keys: exp2(-inf) = 0
                    const float pe = __builtin_amdgcn_exp2f(fmaf(sacc[kt][e], sl2, nmx));
                    sacc[kt][e] = pe;
                    sum += pe;
                }
            sum += __shfl_xor(sum, 32);
            const float inv = 1.f / sum;

            if constexpr (PR) {
                // live attention dropout (vit.py:71, training mode): attn = where(bernoulli(key, keep, (1,H,N,N)), attn / keep, 0)
                // -- word ((h * N + q) * N + key) of the sample's Threefry stream (rng_common.h) decides each probability; the
                // dropped matrix is what multiplies V AND what the block returns.  A run-time branch of the probs variant only.
                if (drop_keys) {
                    const uint32_t k0 = drop_keys[2 * b], k1 = drop_keys[2 * b + 1];
                    const uint32_t nel = (uint32_t)H * N * N, base = ((uint32_t)h * N + (qvalid ? q : 0)) * N;
                    const float rk = 1.f / keep;
                    // the draws first, one key tile per trip of a ROLLED loop into a bit mask (unrolling 16 * NT Threefry calls
                    // next to the score registers goes past the compiler's unroll budget and the scores land in scratch) ...
                    uint64_t mlo = 0, mhi = 0;
#pragma unroll 1
                    for (int kt = 0; kt < NT; ++kt) {
                        uint32_t bits = 0;
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                            const uint32_t el = base + (key < N ? key : 0);
                            bits |= (word_uniform01(stream_word(k0, k1, el, nel)) < keep ? 1u : 0u) << e;
                        }
                        const uint64_t sh = (uint64_t)bits << (16 * (kt & 3));
                        mlo |= kt < 4 ? sh : 0ull;
                        mhi |= kt < 4 ? 0ull : sh;
                    }
                    // ... then applied with static register indices
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const uint64_t m = kt < 4 ? mlo : mhi;
                            sacc[kt][e] = (m >> (16 * (kt & 3) + e)) & 1ull ? sacc[kt][e] * rk : 0.f;
                        }
                }
            }
            if (PR && qvalid && probs) {
                float* pr = probs + (((long long)b * H + h) * N + q) * N;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                        if (kt < NT - 1 || key < N) pr[key] = sacc[kt][e] * inv;
                    }
            }

            // ---- O^T = V^T . P^T  (keys of a k16-step in accumulator-register order; V^T fragments two steps ahead)
            f32x16 oacc[DT];
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
            {
                constexpr int PS = NT * 2, PD = 2;
                uint4 vf[PS][DT];
                auto rdv = [&](int st) {
                    // step st = (key tile kt, half t2): lane half fh owns keys kt*32 + 16*t2 + {4fh..4fh+3, 8+4fh..}
                    // = positions 16*st + 8*fh .. +7 of the permuted row
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        vf[st][dt] = *(const uint4*)(vl + (dt * 32 + fr) * VPITCH + (16 * st + 8 * fh) * 2);
                };
#pragma unroll
                for (int st = 0; st < PD && st < PS; ++st) rdv(st);
#pragma unroll
                for (int st = 0; st < PS; ++st) {
                    const int kt = st >> 1, t2 = st & 1;
                    if (st + PD < PS) rdv(st + PD);
                    uint4 pf;
                    pf.x = pack_bf2(sacc[kt][8 * t2 + 0], sacc[kt][8 * t2 + 1]);
                    pf.y = pack_bf2(sacc[kt][8 * t2 + 2], sacc[kt][8 * t2 + 3]);
                    pf.z = pack_bf2(sacc[kt][8 * t2 + 4], sacc[kt][8 * t2 + 5]);
                    pf.w = pack_bf2(sacc[kt][8 * t2 + 6], sacc[kt][8 * t2 + 7]);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
                        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf[st][dt]),
                                                                           __builtin_bit_cast(bf16x8, pf), oacc[dt], 0,
                                                                           0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (st == PS / 2) {
                        // half of the score registers are dead by now: fetch the Q fragments of the wave's tile of
                        // the NEXT pair; the latency hides behind the rest of P.V, the stores and the LDS fill
                        load_q(qcur, npair, qt);
                    }
                }
            }
            // ---- store: lane = query, 4 consecutive d per accumulator quad ---------------------------
            if (qvalid) {
                bf16_t* orow = out + ((long long)b * N + q) * H * DH + (long long)h * DH;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int d = dt * 32 + 8 * g + 4 * fh;
                        uint2 u;
                        u.x = pack_bf2(oacc[dt][4 * g] * inv, oacc[dt][4 * g + 1] * inv);
                        u.y = pack_bf2(oacc[dt][4 * g + 2] * inv, oacc[dt][4 * g + 3] * inv);
                        *(uint2*)(orow + d) = u;
                    }
            }
        }
        __syncthreads();                         // every wave is done with this pair's K / V^T
    }
}

int mha_mfma_supported(int N, int dh, int dtype) {
    return dtype == MV_BF16 && (dh == 32 || dh == 64) && N >= 1 && N <= 256;
}

template <int DH>
static int mha_launch_dh(const void* qkv, bool hm, void* out, float* probs, int B, int N, int H, float scale,
                         const uint32_t* drop_keys, float keep, hipStream_t st) {
    const int nt = (N + 31) / 32;
    const int pairs = B * H;
    dim3 grid(pairs < 256 ? pairs : 256), block(512);       // persistent: one 8-wave block per CU walks the (image, head) pairs
#define LAUNCH(NT_, HM_, PR_)                                                                           \
    hipLaunchKernelGGL((mha_mfma_kernel<DH, NT_, HM_, PR_>), grid, block, 0, st, (const bf16_t*)qkv, (bf16_t*)out, \
                       probs, B, N, H, scale, drop_keys, keep)
#define GO(NT_)                                                                                                 \
    case NT_:                                                                                                   \
        if (hm && (probs || drop_keys)) LAUNCH(NT_, true, true);                                                \
        else if (hm) LAUNCH(NT_, true, false);                                                                  \
        else if (probs || drop_keys) LAUNCH(NT_, false, true);                                                  \
        else LAUNCH(NT_, false, false);                                                                         \
        break;
    switch (nt) {
        GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8)
        default:
            set_error("mha_mfma: N=%d unsupported", N);
            return MV_E_UNSUPPORTED;
    }
#undef GO
#undef LAUNCH
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mha_mfma_launch(const void* qkv, int head_major, void* out, float* probs, int B, int N, int H, int dh, float scale,
                    const uint32_t* drop_keys, float keep, hipStream_t st) {
    if ((long long)B * H >= (1LL << 31)) {
        set_error("mha_mfma: B*H too large");
        return MV_E_UNSUPPORTED;
    }
    if (head_major) set_kernel_name(dh == 64 ? "mha_mfma_dh64_hm" : "mha_mfma_dh32_hm");
    else set_kernel_name(dh == 64 ? "mha_mfma_dh64" : "mha_mfma_dh32");
    if (dh == 64) return mha_launch_dh<64>(qkv, head_major != 0, out, probs, B, N, H, scale, drop_keys, keep, st);
    return mha_launch_dh<32>(qkv, head_major != 0, out, probs, B, N, H, scale, drop_keys, keep, st);
}

}  // namespace mv

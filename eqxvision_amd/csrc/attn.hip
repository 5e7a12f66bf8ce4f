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

namespace mv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DH, int NT>
__global__ __launch_bounds__(256) void mha_mfma_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                      float* __restrict__ probs, int B, int N, int H, float scale) {
    constexpr int NP = NT * 32;                                     // padded key count
    constexpr int KPITCH = DH * 2 + 16;                             // bytes; odd number of 16-B slots
    constexpr int VPITCH = (NP * 2) + ((((NP * 2) / 8) & 1) ? 0 : 8);  // bytes; 8 * odd
    constexpr int KC = DH / 16;                                     // k16 steps of Q.K^T
    constexpr int DT = DH / 32;                                     // 32-wide d tiles of the output
    __shared__ __attribute__((aligned(16))) char smem[NP * KPITCH + DH * VPITCH];
    char* kl = smem;
    char* vl = smem + NP * KPITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const long long rs = 3LL * H * DH;                               // qkv row stride (elements)
    const bf16_t* qbase = qkv + (long long)b * N * rs + (long long)h * DH;
    const bf16_t* kbase = qbase + (long long)H * DH;
    const bf16_t* vbase = qbase + 2LL * H * DH;

    // ---- stage K: [key][d], 16-byte chunks; loads are batched (4 in flight per thread) and unconditional
    //      (clamped row, zeroed afterwards) so their latencies overlap instead of chaining ------------------
    constexpr int CH = DH / 8;
    constexpr int KITEMS = NP * CH;
    for (int base = 0; base < KITEMS; base += 256 * 4) {
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = base + j * 256 + tid;
            const int key = i / CH, ch = i - key * CH;
            const int kc = key < N ? key : N - 1;
            v[j] = *(const uint4*)(kbase + (long long)kc * rs + ch * 8);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = base + j * 256 + tid;
            const int key = i / CH, ch = i - key * CH;
            if (i < KITEMS) *(uint4*)(kl + key * KPITCH + ch * 16) = key < N ? v[j] : make_uint4(0, 0, 0, 0);
        }
    }
    // ---- stage V^T: [d][key]; each thread transposes a 2-key x 8-d patch into 8 dword stores --------------
    constexpr int VITEMS = (NP / 2) * CH;
    for (int base = 0; base < VITEMS; base += 256 * 2) {
        uint4 v0[2], v1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = base + j * 256 + tid;
            const int ch = i / (NP / 2), kp = i - ch * (NP / 2);      // consecutive lanes -> consecutive key pairs
            const int key = 2 * kp;
            const int k0 = key < N ? key : N - 1, k1 = key + 1 < N ? key + 1 : N - 1;
            const int chc = ch < CH ? ch : CH - 1;
            v0[j] = *(const uint4*)(vbase + (long long)k0 * rs + chc * 8);
            v1[j] = *(const uint4*)(vbase + (long long)k1 * rs + chc * 8);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = base + j * 256 + tid;
            if (i >= VITEMS) continue;
            const int ch = i / (NP / 2), kp = i - ch * (NP / 2);
            const int key = 2 * kp;
            const uint4 z = make_uint4(0, 0, 0, 0);
            const uint4 w0 = key < N ? v0[j] : z, w1 = key + 1 < N ? v1[j] : z;
            const uint32_t a[4] = {w0.x, w0.y, w0.z, w0.w}, c[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t lo = (a[e] & 0xffffu) | (c[e] << 16);          // d = 8ch+2e   : (key, key+1)
                const uint32_t hi = (a[e] >> 16) | (c[e] & 0xffff0000u);      // d = 8ch+2e+1
                *(uint32_t*)(vl + (ch * 8 + 2 * e) * VPITCH + key * 2) = lo;
                *(uint32_t*)(vl + (ch * 8 + 2 * e + 1) * VPITCH + key * 2) = hi;
            }
        }
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    const int nqt = (N + 31) / 32;
    const float sl2 = scale * 1.4426950408889634f;   // exp(scale*x) = exp2(sl2*x)

    for (int qt = wave; qt < nqt; qt += 4) {
        const int q = qt * 32 + fr;
        const bool qvalid = q < N;
        // ---- S^T = K . Q^T -------------------------------------------------------------------
        f32x16 sacc[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc[kt][e] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KC; ++kk) {
            uint4 qf = make_uint4(0, 0, 0, 0);
            if (qvalid) qf = *(const uint4*)(qbase + (long long)q * rs + kk * 16 + fh * 8);
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                const uint4 kf = *(const uint4*)(kl + (kt * 32 + fr) * KPITCH + kk * 32 + fh * 16);
                sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf),
                                                                   __builtin_bit_cast(bf16x8, qf), sacc[kt], 0, 0, 0);
            }
        }
        // ---- softmax over keys: lane = query column; keys = (kt, reg) and the partner half ------
        // accumulator register e of tile kt holds key 32*kt + (e&3) + 8*(e>>2) + 4*fh
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
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = exp2f((sacc[kt][e] - mx) * sl2);   // masked keys: exp2(-inf) = 0
                sacc[kt][e] = pe;
                sum += pe;
            }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.f / sum;

        if (probs && qvalid) {
            float* pr = probs + (((long long)b * H + h) * N + q) * N;
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int key = kt * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                    if (kt < NT - 1 || key < N) pr[key] = sacc[kt][e] * inv;
                }
        }

        // ---- O^T = V^T . P^T  (keys of a k16-step in accumulator-register order) ------------------
        f32x16 oacc[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) oacc[dt][e] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int t2 = 0; t2 < 2; ++t2) {
                uint4 pf;
                pf.x = pack_bf2(sacc[kt][8 * t2 + 0], sacc[kt][8 * t2 + 1]);
                pf.y = pack_bf2(sacc[kt][8 * t2 + 2], sacc[kt][8 * t2 + 3]);
                pf.z = pack_bf2(sacc[kt][8 * t2 + 4], sacc[kt][8 * t2 + 5]);
                pf.w = pack_bf2(sacc[kt][8 * t2 + 6], sacc[kt][8 * t2 + 7]);
                const int key0 = kt * 32 + 16 * t2 + 4 * fh;     // e = 0..3 -> key0 + e ; e = 4..7 -> key0 + 8 + (e-4)
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    const char* vrow = vl + (dt * 32 + fr) * VPITCH + key0 * 2;
                    const uint2 lo = *(const uint2*)(vrow);
                    const uint2 hi = *(const uint2*)(vrow + 16);
                    const uint4 vf = make_uint4(lo.x, lo.y, hi.x, hi.y);
                    oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf),
                                                                       __builtin_bit_cast(bf16x8, pf), oacc[dt], 0,
                                                                       0, 0);
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
}

int mha_mfma_supported(int N, int dh, int dtype) {
    return dtype == MV_BF16 && (dh == 32 || dh == 64) && N >= 1 && N <= 256;
}

template <int DH>
static int mha_launch_dh(const void* qkv, void* out, float* probs, int B, int N, int H, float scale, hipStream_t st) {
    const int nt = (N + 31) / 32;
    dim3 grid(H, B), block(256);
#define GO(NT_)                                                                                                 \
    case NT_:                                                                                                   \
        hipLaunchKernelGGL((mha_mfma_kernel<DH, NT_>), grid, block, 0, st, (const bf16_t*)qkv, (bf16_t*)out, probs, \
                           B, N, H, scale);                                                                     \
        break;
    switch (nt) {
        GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8)
        default:
            set_error("mha_mfma: N=%d unsupported", N);
            return MV_E_UNSUPPORTED;
    }
#undef GO
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mha_mfma_launch(const void* qkv, void* out, float* probs, int B, int N, int H, int dh, float scale,
                    hipStream_t st) {
    if (H > 65535 || B > 65535) {
        set_error("mha_mfma: grid too large");
        return MV_E_UNSUPPORTED;
    }
    set_kernel_name(dh == 64 ? "mha_mfma_dh64" : "mha_mfma_dh32");
    if (dh == 64) return mha_launch_dh<64>(qkv, out, probs, B, N, H, scale, st);
    return mha_launch_dh<32>(qkv, out, probs, B, N, H, scale, st);
}

}  // namespace mv

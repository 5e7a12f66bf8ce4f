// Fully-connected layers over FEW rows (a batch of <= a few hundred feature vectors: alexnet.py:62-70 classifier, 9216 -> 4096 -> 4096
// -> classes; the VGG classifier) on gfx950: y[M][N] = act(x[M][K] . W[N][K]^T + b).  With M = 128 the layer is bound by streaming W
// (75 MB for 9216 x 4096) -- every CU must read its share of W exactly once, at full width, and the x rows it multiplies with
// should reach it as few times as possible:
//   * W arrives pre-arranged by the host in MFMA fragment order ([N / 32 tiles][K / 16 steps][64 lanes][8]): a wave's load of
//     one A fragment is one contiguous 1 KB piece, straight from HBM / L2 into registers (rolling prefetch, 8 steps ahead);
//   * a workgroup (4 waves) owns 128 output columns -- one 32-column tile per wave -- and 128 rows, over a K-RANGE of the
//     reduction (split-K: 32 column blocks x 8 ranges = 256 workgroups for 4096 outputs): the x rows of a 128-wide k-chunk are
//     loaded ONCE per workgroup, coalesced, into LDS (double buffered, the next chunk's global loads fly under the MFMAs) and all
//     four waves read their B fragments from there (row pitch 272 B: conflict-free ds_read_b128); every W fragment feeds 4 MFMAs;
//   * the fp32 partial tiles go to a workspace in accumulator order (coalesced 256-byte stores) and a second small kernel adds
//     the K-ranges IN A FIXED ORDER (bit-reproducible), adds the bias, applies the activation and writes y.
// The un-split kernel this replaces for these shapes (skinny.hip: one 32 x 32 tile per workgroup, W rows gathered 32 bytes at a
// time) ran 9216 -> 4096 at M = 128 in 80 us = 0.96 TB/s of weight bytes.
#include "mfma_common.h"

namespace mv {

namespace {

struct FcP {
    const bf16_t* x;      // [M][K]
    const bf16_t* wf;     // [NT][K / 16][64][8]
    float* ws;            // [S][NT][MT][16][64] fp32 partial accumulators
    long long M;
    int K, NT, S, NC;     // NT = column tiles of 32 (N rounded up); S k-ranges; NC = K / 128 chunks
};

constexpr int FC_CH = 128;                 // k-chunk
constexpr int FC_PITCH = FC_CH * 2 + 16;   // LDS row pitch of a chunk (bytes)
constexpr int FC_MB = 128;                 // rows per workgroup

__global__ __launch_bounds__(256) void fc_stream_kernel(const FcP p) {
    constexpr int D = 8, KSC = FC_CH / 16;                      // prefetch depth; k-steps per chunk
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int nt = blockIdx.x * 4 + wave;                        // my column tile (may be past NT: clamped, result dropped)
    const int s = blockIdx.y;
    const long long m0 = (long long)blockIdx.z * FC_MB;
    const int c0 = (int)((long long)p.NC * s / p.S), c1 = (int)((long long)p.NC * (s + 1) / p.S);
    const int nsteps = (c1 - c0) * KSC;
    const int ntc = nt < p.NT ? nt : p.NT - 1;
    const uint4* wp = (const uint4*)p.wf + ((size_t)ntc * (p.K / 16) + (size_t)c0 * KSC) * 64 + lane;
    uint4 a[D];
#pragma unroll
    for (int d = 0; d < D; ++d) a[d] = wp[(size_t)(d < nsteps ? d : (nsteps > 0 ? nsteps - 1 : 0)) * 64];

    // x chunk loader: 128 rows x 256 bytes = 2048 pieces of 16 bytes, 8 per thread; rows past M read row M - 1 and are zeroed
    const int lr = tid >> 4, lc = tid & 15;                      // piece i of the thread: row lr + 16 i, 16-byte column lc
    uint4 xr[8];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const long long m = m0 + lr + 16 * i;
            const bool ok = m < p.M;
            const uint4 v = *(const uint4*)(p.x + (ok ? m : p.M - 1) * p.K + (size_t)c * FC_CH + lc * 8);
            xr[i] = ok ? v : make_uint4(0, 0, 0, 0);
        }
    };
    auto store_chunk = [&](int buf) {
        char* b = smem + buf * (FC_MB * FC_PITCH);
#pragma unroll
        for (int i = 0; i < 8; ++i) *(uint4*)(b + (lr + 16 * i) * FC_PITCH + lc * 16) = xr[i];
    };

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    if (c0 < c1) {
        load_chunk(c0);
        store_chunk(0);
    }
    __syncthreads();
    int step = 0;
    for (int c = c0; c < c1; ++c) {
        const int buf = (c - c0) & 1;
        const bool more = c + 1 < c1;
        if (more) load_chunk(c + 1);                            // lands under the 32 MFMAs below
        const char* xb = smem + buf * (FC_MB * FC_PITCH) + fr * FC_PITCH + fh * 16;
#pragma unroll
        for (int j = 0; j < KSC; ++j) {
            int nx = step + j + D;
            nx = nx < nsteps ? nx : nsteps - 1;
            const bf16x8 af = __builtin_bit_cast(bf16x8, a[j % D]);
            a[j % D] = wp[(size_t)nx * 64];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bf16x8 bf_ = *(const bf16x8*)(xb + t * 32 * FC_PITCH + j * 32);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf_, acc[t], 0, 0, 0);
            }
        }
        step += KSC;
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }
    if (nt < p.NT) {                                            // accumulator order: [e][lane], 256-byte rows
        float* o = p.ws + ((((size_t)blockIdx.z * p.S + s) * p.NT + nt) * 4) * 1024 + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[(size_t)t * 1024 + e * 64] = acc[t][e];
    }
}

struct FcRP {
    const float* ws;
    const float* bias;    // [N] or nullptr
    void* y;              // [M][N] bf16 or fp32
    long long M;
    int N, NT, S, act, out_f32;
};

// one thread = 4 consecutive output columns of one row: accumulator registers 4 g .. 4 g + 3 of lane (row, half)
__global__ __launch_bounds__(256) void fc_reduce_kernel(const FcRP p) {
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int nt = blockIdx.x, mt = blockIdx.y & 3, mb = blockIdx.y >> 2;
    const long long m = (long long)mb * FC_MB + 32 * mt + (lane & 31);
    const int n = 32 * nt + 8 * g + 4 * (lane >> 5);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.S; ++s) {                              // fixed order: bit-reproducible
        const float* o = p.ws + ((((size_t)mb * p.S + s) * p.NT + nt) * 4 + mt) * 1024 + (4 * g) * 64 + lane;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += o[e * 64];
    }
    if (m >= p.M) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (p.bias && n + e < p.N) v[e] += p.bias[n + e];
        v[e] = apply_act_rt(v[e], p.act);
    }
    if (n + 3 < p.N) {
        if (p.out_f32) *(float4*)((float*)p.y + m * p.N + n) = make_float4(v[0], v[1], v[2], v[3]);
        else {
            uint2 u;
            u.x = pack_bf2(v[0], v[1]);
            u.y = pack_bf2(v[2], v[3]);
            *(uint2*)((bf16_t*)p.y + m * p.N + n) = u;
        }
    } else {
        for (int e = 0; e < 4 && n + e < p.N; ++e) {
            if (p.out_f32) ((float*)p.y)[m * p.N + n + e] = v[e];
            else ((bf16_t*)p.y)[m * p.N + n + e] = f2bf(v[e]);
        }
    }
}

static int fc_splits(long long M, int N, int K) {
    const int nb = (N + 127) / 128, nc = K / FC_CH, mb = (int)((M + FC_MB - 1) / FC_MB);
    int s = 512 / (nb * mb);                                     // ~two workgroups per CU (70 KB of LDS each)
    if (s > nc) s = nc;
    if (s > K / 512) s = K / 512;                                // partial sums written + read (8 S bytes per output of 128 rows) <= the weight bytes
    if (s > 32) s = 32;
    return s < 1 ? 1 : s;
}

}  // namespace

}  // namespace mv

extern "C" {

int mv_fc_stream_supported(int64_t M, int N, int K, int in_dtype, int out_dtype) {
    if (mv::get_flag("no_fc_stream") || mv::get_flag("force_generic")) return 0;
    return in_dtype == MV_BF16 && (out_dtype == MV_BF16 || out_dtype == MV_F32) && M >= 1 && M <= 512 && K % mv::FC_CH == 0 && K >= 512 &&
           N >= 256 && N % 4 == 0 && (long long)N * K >= (1LL << 21);
}

int64_t mv_fc_stream_workspace(int64_t M, int N, int K) {
    const long long nt = (N + 31) / 32, mb = (M + mv::FC_MB - 1) / mv::FC_MB;
    return mb * mv::fc_splits(M, N, K) * nt * 4 * 1024 * 4;
}

int mv_fc_stream_fwd(const void* x, const void* w_frag, const float* bias, void* y, void* workspace, int64_t workspace_bytes, int64_t M,
                     int N, int K, int act, int in_dtype, int out_dtype, mv_stream_t stream_) {
    using namespace mv;
    hipStream_t stream = (hipStream_t)stream_;
    MV_CHECK_ARG(x && w_frag && y && workspace, "mv_fc_stream_fwd: null argument");
    if (!mv_fc_stream_supported(M, N, K, in_dtype, out_dtype)) {
        set_error("mv_fc_stream_fwd: unsupported M=%lld N=%d K=%d (ask mv_fc_stream_supported first)", (long long)M, N, K);
        return MV_E_UNSUPPORTED;
    }
    MV_CHECK_ARG(workspace_bytes >= mv_fc_stream_workspace(M, N, K), "mv_fc_stream_fwd: workspace of %lld bytes, need %lld",
                 (long long)workspace_bytes, (long long)mv_fc_stream_workspace(M, N, K));
    MV_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w_frag & 15) == 0 && ((uintptr_t)y & 15) == 0, "mv_fc_stream_fwd: 16-byte alignment");
    FcP p;
    p.x = (const bf16_t*)x; p.wf = (const bf16_t*)w_frag; p.ws = (float*)workspace; p.M = M; p.K = K;
    p.NT = (N + 31) / 32; p.S = fc_splits(M, N, K); p.NC = K / FC_CH;
    const int mb = (int)((M + FC_MB - 1) / FC_MB);
    constexpr int SMEM = 2 * FC_MB * FC_PITCH;
    MV_HIP(hipFuncSetAttribute((const void*)fc_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM));
    set_kernel_name("fc_stream_bf16");
    hipLaunchKernelGGL(fc_stream_kernel, dim3((unsigned)((p.NT + 3) / 4), (unsigned)p.S, (unsigned)mb), dim3(256), SMEM, stream, p);
    MV_LAUNCH_CHECK();
    FcRP r;
    r.ws = p.ws; r.bias = bias; r.y = y; r.M = M; r.N = N; r.NT = p.NT; r.S = p.S; r.act = act; r.out_f32 = out_dtype == MV_F32;
    hipLaunchKernelGGL(fc_reduce_kernel, dim3((unsigned)p.NT, (unsigned)(4 * mb)), dim3(256), 0, stream, r);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

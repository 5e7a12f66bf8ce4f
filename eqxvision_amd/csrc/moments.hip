// Per-channel batch moments of an NHWC map / row matrix: the statistics half of eqx.experimental.BatchNorm's TRAINING branch
// (called from resnet.py:132-136, 252, 301 when the model is not in inference mode):
//   batch_mean[c] = pmean(mean(x[c]));  batch_var[c] = pmean(mean((x[c] - batch_mean[c])^2))      (two passes, like the reference)
// One launch produces out[c] = sum_rows (x[r][c] - shift[c])^p, p = 1 (shift = NULL) or 2; the caller divides by the global row
// count after the cross-rank sum (mv_allreduce_sum_f32).  HBM-bound: rows * C * sizeof(T) bytes read once per pass.
// Deterministic: block b reduces rows b, b + G, ... into partial[b][C] (fixed order), a second kernel adds the G partials in a fixed order.
#include "mfma_common.h"

namespace mv {

constexpr int MOM_BLOCKS = 1024;                           // 4 blocks of 4 waves per CU: enough loads in flight to stream HBM

template <typename T>
__global__ __launch_bounds__(256) void moments_partial_kernel(const T* x, const float* shift, float* partial, long long rows, int C,
                                                              int sq) {
    extern __shared__ float red[];                          // [RS][C]
    const int V = C >> 3;                                   // 8-channel vectors per row
    const int RS = 256 / V;                                 // rows a block covers per step
    const int rs = threadIdx.x / V, v = threadIdx.x - rs * V;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rs < RS) {
        float sh[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (shift) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sh[e] = shift[v * 8 + e];
        }
        const long long step = (long long)gridDim.x * RS;
        long long r = (long long)blockIdx.x * RS + rs;
        auto accum = [&](const float4& a, const float4& b) {
            const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = xv[e] - sh[e];
                acc[e] += sq ? d * d : d;
            }
        };
        for (; r + 3 * step < rows; r += 4 * step) {        // four rows in flight per thread (the summation order stays fixed)
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T* p = x + (r + u * step) * C + v * 8;
                a[u] = Out4<T>::ld(p);
                b[u] = Out4<T>::ld(p + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) accum(a[u], b[u]);
        }
        for (; r < rows; r += step) {
            const T* p = x + r * C + v * 8;
            accum(Out4<T>::ld(p), Out4<T>::ld(p + 4));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rs * C + v * 8 + e] = acc[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int j = 0; j < RS; ++j) s += red[j * C + c];
        partial[(long long)blockIdx.x * C + c] = s;
    }
}

// Both moments in ONE pass about a shift that is close to the mean (the running mean of the previous steps -- identical on every
// rank, so the sums of the ranks add up): partial[b][0:C] = sum (x - shift), partial[b][C:2C] = sum (x - shift)^2.
// mean = shift + S1 / n, var = S2 / n - (S1 / n)^2: no cancellation worth speaking of when |mean - shift| is a fraction of sigma.
template <typename T>
__global__ __launch_bounds__(256) void moments2_partial_kernel(const T* x, const float* shift, float* partial, long long rows, int C) {
    extern __shared__ float red[];                          // [RS][2C]
    const int V = C >> 3;
    const int RS = 256 / V;
    const int rs = threadIdx.x / V, v = threadIdx.x - rs * V;
    if (rs < RS) {
        float sh[8], a1[8], a2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sh[e] = shift[v * 8 + e];
            a1[e] = 0.f;
            a2[e] = 0.f;
        }
        const long long step = (long long)gridDim.x * RS;
        long long r = (long long)blockIdx.x * RS + rs;
        auto accum = [&](const float4& a, const float4& b) {
            const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = xv[e] - sh[e];
                a1[e] += d;
                a2[e] = fmaf(d, d, a2[e]);
            }
        };
        for (; r + 3 * step < rows; r += 4 * step) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const T* p = x + (r + u * step) * C + v * 8;
                a[u] = Out4<T>::ld(p);
                b[u] = Out4<T>::ld(p + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) accum(a[u], b[u]);
        }
        for (; r < rows; r += step) {
            const T* p = x + r * C + v * 8;
            accum(Out4<T>::ld(p), Out4<T>::ld(p + 4));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[rs * 2 * C + v * 8 + e] = a1[e];
            red[rs * 2 * C + C + v * 8 + e] = a2[e];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * C; c += 256) {
        float s = 0.f;
        for (int j = 0; j < RS; ++j) s += red[j * 2 * C + c];
        partial[(long long)blockIdx.x * 2 * C + c] = s;
    }
}

// out[c] = sum over the G block partials, always in the same order: 16 slices of the partials per channel, then the 16 slice sums
__global__ __launch_bounds__(1024) void moments_final_kernel(const float* partial, float* out, int G, int C) {
    __shared__ float red[16][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C) {
        int g = q;
        for (; g + 48 < G; g += 64) {                       // four partials in flight; added in the order they would be one by one
            const float a0 = partial[(long long)g * C + c], a1 = partial[(long long)(g + 16) * C + c];
            const float a2 = partial[(long long)(g + 32) * C + c], a3 = partial[(long long)(g + 48) * C + c];
            s = (((s + a0) + a1) + a2) + a3;
        }
        for (; g < G; g += 16) s += partial[(long long)g * C + c];
    }
    red[q][cl] = s;
    __syncthreads();
    if (q == 0 && c < C) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) t += red[j][cl];
        out[c] = t;
    }
}

// mean[c] = sum[c] / n, n from the host (one rank) or from the device (the all-reduced row count of ragged shards)
__global__ void bn_mean_kernel(const float* sum, const float* count_dev, float count_host, float* mean, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) mean[c] = sum[c] / (count_dev ? count_dev[0] : count_host);
}

// The rest of the training-mode update, on the device (no host round trip between the launches of a step):
//   var = sqdev / n;  first call: running = (mean, var), else running = (1 - momentum) * batch + momentum * running (in place);
//   scale = weight / sqrt(running_var + eps), shift = bias - running_mean * scale      -- what the normalisation launch applies
__global__ void bn_ema_fold_kernel(const float* sqdev, const float* mean, const float* count_dev, float count_host, float* run_mean,
                                   float* run_var, const float* weight, const float* bias, float* scale, float* shift, float momentum,
                                   float eps, int first_time, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float var = sqdev[c] / (count_dev ? count_dev[0] : count_host);
    float rm = mean[c], rv = var;
    if (!first_time) {
        rm = (1.f - momentum) * rm + momentum * run_mean[c];
        rv = (1.f - momentum) * rv + momentum * run_var[c];
    }
    run_mean[c] = rm;
    run_var[c] = rv;
    const float inv = 1.0f / sqrtf(rv + eps);
    const float sc = (weight ? weight[c] : 1.f) * inv;
    scale[c] = sc;
    shift[c] = (bias ? bias[c] : 0.f) - rm * sc;
}

// single-pass form: sums[0:C] = sum (x - shift), sums[C:2C] = sum (x - shift)^2 over the global batch, shift = the running mean the
// pass was centred on (== run_mean on entry)
__global__ void bn_ema_fold1_kernel(const float* sums, const float* count_dev, float count_host, float* run_mean, float* run_var,
                                    const float* weight, const float* bias, float* scale, float* shift, float momentum, float eps, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float n = count_dev ? count_dev[0] : count_host;
    const float d = sums[c] / n;
    const float var = fmaxf(sums[C + c] / n - d * d, 0.f);
    const float mean = run_mean[c] + d;
    const float rm = (1.f - momentum) * mean + momentum * run_mean[c];
    const float rv = (1.f - momentum) * var + momentum * run_var[c];
    run_mean[c] = rm;
    run_var[c] = rv;
    const float inv = 1.0f / sqrtf(rv + eps);
    const float sc = (weight ? weight[c] : 1.f) * inv;
    scale[c] = sc;
    shift[c] = (bias ? bias[c] : 0.f) - rm * sc;
}

}  // namespace mv

using namespace mv;

extern "C" {

// floats of workspace mv_channel_moments_fwd needs for C channels
int mv_channel_moments_ws(int C) { return MOM_BLOCKS * C; }

int mv_channel_moments_supported(int64_t rows, int C, int dtype) {
    return (dtype == MV_BF16 || dtype == MV_F32) && rows > 0 && C > 0 && C % 8 == 0 && C <= 2048;
}

int mv_channel_moments_fwd(const void* x, const float* shift, float* out, float* workspace, int64_t rows, int C, int squared,
                           int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && out && workspace, "channel_moments: NULL pointer");
    if (!mv_channel_moments_supported(rows, C, dtype)) {
        set_error("channel_moments: unsupported rows=%lld C=%d dtype=%d (C must be a multiple of 8, <= 2048)", (long long)rows, C, dtype);
        return MV_E_UNSUPPORTED;
    }
    const int RS = 256 / (C >> 3);
    long long need = (rows + 16LL * RS - 1) / (16LL * RS);   // >= 16 row steps per block: the G x C partials stay small next to x
    const int G = (int)(need < MOM_BLOCKS ? need : MOM_BLOCKS);
    const size_t smem = (size_t)RS * C * sizeof(float);
    set_kernel_name(squared ? "channel_sqdev" : "channel_sum");
    if (dtype == MV_F32)
        hipLaunchKernelGGL(moments_partial_kernel<float>, dim3(G), dim3(256), smem, (hipStream_t)stream, (const float*)x, shift, workspace,
                           (long long)rows, C, squared);
    else
        hipLaunchKernelGGL(moments_partial_kernel<bf16_t>, dim3(G), dim3(256), smem, (hipStream_t)stream, (const bf16_t*)x, shift, workspace,
                           (long long)rows, C, squared);
    MV_LAUNCH_CHECK();
    hipLaunchKernelGGL(moments_final_kernel, dim3((C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, workspace, out, G, C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

// workspace: 2 * mv_channel_moments_ws(C) floats; out: 2 * C floats
int mv_channel_moments2_fwd(const void* x, const float* shift, float* out, float* workspace, int64_t rows, int C, int dtype,
                            mv_stream_t stream) {
    MV_CHECK_ARG(x && shift && out && workspace, "channel_moments2: NULL pointer");
    if (!mv_channel_moments_supported(rows, C, dtype)) {
        set_error("channel_moments2: unsupported rows=%lld C=%d dtype=%d (C must be a multiple of 8, <= 2048)", (long long)rows, C, dtype);
        return MV_E_UNSUPPORTED;
    }
    const int RS = 256 / (C >> 3);
    long long need = (rows + 16LL * RS - 1) / (16LL * RS);
    const int G = (int)(need < MOM_BLOCKS ? need : MOM_BLOCKS);
    const size_t smem = (size_t)RS * 2 * C * sizeof(float);
    set_kernel_name("channel_moments2");
    if (dtype == MV_F32)
        hipLaunchKernelGGL(moments2_partial_kernel<float>, dim3(G), dim3(256), smem, (hipStream_t)stream, (const float*)x, shift, workspace,
                           (long long)rows, C);
    else
        hipLaunchKernelGGL(moments2_partial_kernel<bf16_t>, dim3(G), dim3(256), smem, (hipStream_t)stream, (const bf16_t*)x, shift,
                           workspace, (long long)rows, C);
    MV_LAUNCH_CHECK();
    hipLaunchKernelGGL(moments_final_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, (hipStream_t)stream, workspace, out, G, 2 * C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_bn_ema_fold1_fwd(const float* sums, const float* count_dev, float count_host, float* run_mean, float* run_var,
                        const float* weight, const float* bias, float* scale, float* shift, float momentum, float eps, int C,
                        mv_stream_t stream) {
    MV_CHECK_ARG(sums && run_mean && run_var && scale && shift && C > 0 && (count_dev || count_host > 0.f), "bn_ema_fold1: bad arguments");
    set_kernel_name("bn_ema_fold1");
    hipLaunchKernelGGL(bn_ema_fold1_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count_dev, count_host, run_mean,
                       run_var, weight, bias, scale, shift, momentum, eps, C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_bn_mean_fwd(const float* sum, const float* count_dev, float count_host, float* mean, int C, mv_stream_t stream) {
    MV_CHECK_ARG(sum && mean && C > 0 && (count_dev || count_host > 0.f), "bn_mean: bad arguments");
    set_kernel_name("bn_mean");
    hipLaunchKernelGGL(bn_mean_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sum, count_dev, count_host, mean, C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_bn_ema_fold_fwd(const float* sqdev, const float* mean, const float* count_dev, float count_host, float* run_mean,
                       float* run_var, const float* weight, const float* bias, float* scale, float* shift, float momentum, float eps,
                       int first_time, int C, mv_stream_t stream) {
    MV_CHECK_ARG(sqdev && mean && run_mean && run_var && scale && shift && C > 0 && (count_dev || count_host > 0.f),
                 "bn_ema_fold: bad arguments");
    set_kernel_name("bn_ema_fold");
    hipLaunchKernelGGL(bn_ema_fold_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sqdev, mean, count_dev, count_host,
                       run_mean, run_var, weight, bias, scale, shift, momentum, eps, first_time, C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

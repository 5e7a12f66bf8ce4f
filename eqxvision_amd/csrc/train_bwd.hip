// Backward half of a training step (reference tests/test_grads.py:35-47: eqx.filter_value_and_grad +
// optax.adam on the classification models): the gradient kernels that have no forward twin, all fp32.  The contractions that
// ARE a forward contraction with other operands (Linear dgrad / wgrad, attention's four products) go through the forward
// entries (mv_linear_fwd on transposed operands, eqxvision_amd/grad.py); what lives here:
//   * Conv2d dgrad / wgrad for any filter, stride, padding, dilation (groups = 1), NHWC maps, KRSC filters;
//   * activation (ReLU / GELU-tanh), max-pool and global-average-pool backward;
//   * column reductions (bias, BatchNorm / LayerNorm gamma and beta gradients), the BatchNorm gamma gradient for a normalisation
//     with RUNNING statistics (the reference's training branch normalises with the updated running statistics, which the
//     gradient treats as constants: eqx.experimental.BatchNorm keeps its state outside the differentiated pytree);
//   * LayerNorm and softmax backward, softmax cross-entropy with its gradient, the Adam update, a 2-D transpose.
// Minimum slice: correctness first (one thread per output element, VALU); the matrix-core versions are future work.
#include "common.h"
#include "mfma_common.h"

namespace mv {

namespace {

// groups: input channel c belongs to group c / (C / groups), whose K / groups filters hold C / groups channels each (KRSC rows)
__global__ void conv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int N, int H,
                                  int W, int C, int K, int R, int S, int Ho, int Wo, int sh, int sw, int ph, int pw, int dh,
                                  int dw, int groups) {
    const int Cg = C / groups, Kg = K / groups;
    const long long pix = blockIdx.x;                       // (n, hi, wi)
    const int wi = (int)(pix % W), hi = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int r = 0; r < R; ++r) {
            const int th = hi + ph - r * dh;
            if (th < 0 || th % sh) continue;
            const int ho = th / sh;
            if (ho >= Ho) continue;
            for (int s = 0; s < S; ++s) {
                const int tw = wi + pw - s * dw;
                if (tw < 0 || tw % sw) continue;
                const int wo = tw / sw;
                if (wo >= Wo) continue;
                const int gi = c / Cg;
                const float* dyp = dy + (((long long)n * Ho + ho) * Wo + wo) * K + gi * Kg;
                const float* wp = w + ((long long)gi * Kg * R * S + (long long)r * S + s) * Cg + (c - gi * Cg);
                for (int k = 0; k < Kg; ++k) acc = fmaf(dyp[k], wp[(long long)k * R * S * Cg], acc);
            }
        }
        dx[pix * C + c] = acc;
    }
}

// one block per (k, r, s); threads over c; the block's threads walk the output positions together
// Input gradient on the fp32 matrix cores (groups = 1): dx[pos][c] = sum over (r, s, k) of dy[out(pos, r, s)][k] * w[k][r][s][c].  One wave =
// 32 input channels x 32 input positions; A = weights (lane: channel c0 + fr, coalesced along c; four k per step group), B = dy rows
// (a lane loads four consecutive k of its position's output pixel); a tap contributes to a position only where the stride divides.
__global__ __launch_bounds__(256) void conv_dgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                              int N, int H, int W, int C, int K, int R, int S, int Ho, int Wo, int sh,
                                                              int sw, int ph, int pw, int dh, int dw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 31, fh = lane >> 5;
    const long long M = (long long)N * H * W;
    const long long m = ((long long)blockIdx.x * 4 + wave) * 32 + fr;
    const int c0 = blockIdx.y * 32;
    const bool mok = m < M;
    const long long mm = mok ? m : 0;
    const int wi = (int)(mm % W), hi = (int)((mm / W) % H), n = (int)(mm / ((long long)W * H));
    const bool cok = c0 + fr < C;
    const float* wc = w + (cok ? c0 + fr : 0);
    const long long wks = (long long)R * S * C;             // stride between filters
    const bool vec = (K & 3) == 0;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int r = 0; r < R; ++r) {
        const int th = hi + ph - r * dh;
        const int ho = th / sh;
        const bool hok = th >= 0 && th - ho * sh == 0 && ho < Ho;
        for (int s_ = 0; s_ < S; ++s_) {
            const int tw = wi + pw - s_ * dw;
            const int wo = tw / sw;
            const bool inside = mok && hok && tw >= 0 && tw - wo * sw == 0 && wo < Wo;
            const float* dyp = dy + (inside ? (((long long)n * Ho + ho) * Wo + wo) * K : 0);
            const float* wp = wc + ((long long)r * S + s_) * C;
            if (vec) {
                for (int k0 = 0; k0 < K; k0 += 8) {
                    const int k = k0 + 4 * fh;
                    const bool kin = k < K;
                    const float4 b = (inside && kin) ? *(const float4*)(dyp + k) : make_float4(0.f, 0.f, 0.f, 0.f);
                    float a[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = (cok && kin) ? wp[(long long)(k + e) * wks] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b.w, acc, 0, 0, 0);
                }
            } else {
                for (int k0 = 0; k0 < K; k0 += 2) {
                    const int k = k0 + fh;
                    const bool kin = k < K;
                    const float a = (cok && kin) ? wp[(long long)k * wks] : 0.f;
                    const float b = (inside && kin) ? dyp[k] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
        }
    }
    if (!mok) return;
    // acc[4 q + i]: channel c0 + 8 q + 4 fh + i of position fr
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = c0 + 8 * q + 4 * fh + i;
            if (c < C) dx[m * C + c] = acc[4 * q + i];
        }
}

__global__ void conv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dwt, int N, int H,
                                  int W, int Cfull, int K, int R, int S, int Ho, int Wo, int sh, int sw, int ph, int pw, int dh,
                                  int dw, int groups) {
    const int krs = blockIdx.x;
    const int s = krs % S, r = (krs / S) % R, k = krs / (R * S);
    const int C = Cfull / groups;                          // channels per filter; the filter's group starts at channel c0
    const int c0 = (k / (K / groups)) * C;
    x += c0;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n)
            for (int ho = 0; ho < Ho; ++ho) {
                const int hi = ho * sh - ph + r * dh;
                if ((unsigned)hi >= (unsigned)H) continue;
                for (int wo = 0; wo < Wo; ++wo) {
                    const int wi = wo * sw - pw + s * dw;
                    if ((unsigned)wi >= (unsigned)W) continue;
                    acc = fmaf(dy[(((long long)n * Ho + ho) * Wo + wo) * K + k], x[(((long long)n * H + hi) * W + wi) * Cfull + c], acc);
                }
            }
        dwt[(long long)krs * C + c] = acc;
    }
}

__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ ref, float* __restrict__ dx, long long n,
                               int act) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float g = dy[i], v = ref[i];
        float d = g;
        if (act == MV_ACT_RELU) {
            d = v > 0.f ? g : 0.f;
        } else if (act == MV_ACT_GELU_TANH) {                // d/dx [0.5 x (1 + tanh(u))], u = c (x + 0.044715 x^3)
            const float c0 = 0.7978845608028654f, c1 = 0.044715f;
            const float u = c0 * (v + c1 * v * v * v);
            const float t = tanhf(u);
            d = g * (0.5f * (1.f + t) + 0.5f * v * (1.f - t * t) * c0 * (1.f + 3.f * c1 * v * v));
        } else if (act == MV_ACT_HARD_SIGMOID) {             // clip(x + 3, 0, 6) / 6
            d = (v > -3.f && v < 3.f) ? g * (1.0f / 6.0f) : 0.f;
        } else if (act == MV_ACT_HARD_SWISH) {               // x clip(x + 3, 0, 6) / 6
            d = v <= -3.f ? 0.f : (v >= 3.f ? g : g * (2.f * v + 3.f) * (1.0f / 6.0f));
        } else if (act == MV_ACT_SIGMOID || act == MV_ACT_SILU) {
            const float sg = 1.0f / (1.0f + expf(-v));
            d = act == MV_ACT_SIGMOID ? g * sg * (1.f - sg) : g * sg * (1.f + v * (1.f - sg));
        }
        dx[i] = d;
    }
}

// gather form: an input element receives dy of every window in which it is the FIRST maximum (row-major scan of the window,
// like the forward's max); no atomics, deterministic
__global__ void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int N, int H,
                                   int W, int C, int kh, int kw, int sh, int sw, int ph, int pw, int Ho, int Wo) {
    const long long pix = blockIdx.x;
    const int wi = (int)(pix % W), hi = (int)((pix / W) % H), n = (int)(pix / ((long long)W * H));
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float me = x[pix * C + c];
        float acc = 0.f;
        for (int ho = 0; ho < Ho; ++ho) {
            const int h0 = ho * sh - ph;
            if (hi < h0 || hi >= h0 + kh) continue;
            for (int wo = 0; wo < Wo; ++wo) {
                const int w0 = wo * sw - pw;
                if (wi < w0 || wi >= w0 + kw) continue;
                bool first = true;                            // am I the first maximum of window (ho, wo)?
                for (int r = 0; r < kh && first; ++r)
                    for (int s = 0; s < kw; ++s) {
                        const int a = h0 + r, b = w0 + s;
                        if ((unsigned)a >= (unsigned)H || (unsigned)b >= (unsigned)W) continue;
                        const float v = x[(((long long)n * H + a) * W + b) * C + c];
                        const bool before = a < hi || (a == hi && b < wi);
                        if (v > me || (before && v == me)) { first = false; break; }
                    }
                if (first) acc += dy[(((long long)n * Ho + ho) * Wo + wo) * C + c];
            }
        }
        dx[pix * C + c] = acc;
    }
}

__global__ void avgpool_global_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int HW, int C, long long n) {
    const float inv = 1.0f / (float)HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long long b = i / ((long long)C * HW);
        dx[i] = dy[b * C + c] * inv;
    }
}

// out[c] = sum_m a[m, c] * (b ? b[m, c] : 1): one block per 64 columns, 4 row groups reduced through LDS
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                      long long M, int C) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float acc = 0.f;
    if (c < C)
        for (long long m = rg; m < M; m += 4) acc += b ? a[m * C + c] * b[m * C + c] : a[m * C + c];
    part[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < C) out[c] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// the same sum with the rows split over blockIdx.y (chunk rows each): part[y][c]; colsum_finish adds the parts in a fixed order
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part,
                                                           long long M, int C, long long chunk) {
    __shared__ float sh[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const long long m0 = blockIdx.y * chunk, m1 = m0 + chunk < M ? m0 + chunk : M;
    float acc = 0.f;
    if (c < C)
        for (long long m = m0 + rg; m < m1; m += 4) acc += b ? a[m * C + c] * b[m * C + c] : a[m * C + c];
    sh[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < C) part[(long long)blockIdx.y * C + c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}
// out[i] = part[0][i] + part[1][i] + ... (fixed order: bit-reproducible)
__global__ void sum_parts_kernel(const float* __restrict__ part, float* __restrict__ out, long long n, int parts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    for (int s = 0; s < parts; ++s) acc += part[(long long)s * n + i];
    out[i] = acc;
}

// Weight gradient on the fp32 matrix cores: for one filter tap (r, s), dW[k][c] = sum over output positions of dy[pos][k] * x[pos @ (r, s)][c]
// is a GEMM whose reduction runs over the positions.  One wave = one 32 (k) x 32 (c) tile of one tap over a chunk of positions,
// v_mfma_f32_32x32x2_f32 reduces two positions per instruction: lane (fr, fh) feeds dy[pos + fh][k0 + fr] and x[..][c0 + fr] -- both
// coalesced along the channel index, no LDS.  The chunks' partial tiles are added in a fixed order by sum_parts_kernel.
__global__ __launch_bounds__(64) void conv_wgrad_mfma_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                             int N, int H, int W, int Cfull, int K, int R, int S, int Ho, int Wo, int sh,
                                                             int sw, int ph, int pw, int dh, int dw, int groups, long long P,
                                                             long long chunk, int ktiles, int ctiles) {
    const int lane = threadIdx.x, fr = lane & 31, fh = lane >> 5;
    int t = blockIdx.x;
    const int s = t % S; t /= S;
    const int r = t % R; t /= R;
    const int ct = t % ctiles; t /= ctiles;
    const int kt = t % ktiles;
    const int g = t / ktiles;
    const int Kg = K / groups, Cg = Cfull / groups;
    const int kl = kt * 32 + fr, cl = ct * 32 + fr;
    const bool kok = kl < Kg, cok = cl < Cg;
    const float* dyk = dy + (g * Kg + (kok ? kl : 0));
    const float* xc = x + (g * Cg + (cok ? cl : 0));
    const long long p0 = blockIdx.y * chunk, p1 = p0 + chunk < P ? p0 + chunk : P;
    long long p = p0 + fh;
    int n = (int)(p / ((long long)Ho * Wo));
    int rem = (int)(p - (long long)n * Ho * Wo);
    int ho = rem / Wo, wo = rem - ho * Wo;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (; p - fh < p1; p += 2) {
        const bool in = p < p1;
        const int hi = ho * sh - ph + r * dh, wi = wo * sw - pw + s * dw;
        const bool inside = in && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
        const float a = (in && kok) ? dyk[p * K] : 0.f;
        const float b = (inside && cok) ? xc[(((long long)n * H + hi) * W + wi) * Cfull] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        wo += 2;
        while (wo >= Wo) {
            wo -= Wo;
            if (++ho >= Ho) { ho = 0; ++n; }
        }
    }
    // acc[i]: k = 8 (i / 4) + 4 fh + i % 4, c = fr
    float* o = part + (long long)blockIdx.y * ((long long)K * R * S * Cg);
    if (cok) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = kt * 32 + 8 * (i >> 2) + 4 * fh + (i & 3);
            if (kk < Kg) o[(((long long)(g * Kg + kk) * R + r) * S + s) * Cg + cl] = acc[i];
        }
    }
}

// ds[b, c] = sum over the HW positions of image b of g[b, p, c] * x[b, p, c]  (the scale's gradient of y = x * s[b, c])
__global__ __launch_bounds__(256) void channel_scale_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                 float* __restrict__ ds, int HW, int C) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, b = blockIdx.y;
    float acc = 0.f;
    if (c < C)
        for (int p = rg; p < HW; p += 4) acc = fmaf(g[((long long)b * HW + p) * C + c], x[((long long)b * HW + p) * C + c], acc);
    part[rg][cl] = acc;
    __syncthreads();
    if (rg == 0 && c < C) ds[(long long)b * C + c] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

__global__ void bn_dgamma_kernel(const float* __restrict__ dyz, const float* __restrict__ dys, const float* __restrict__ mean,
                                 const float* __restrict__ var, float eps, float* __restrict__ dgamma, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) dgamma[c] = (dyz[c] - mean[c] * dys[c]) * rsqrtf(var[c] + eps);        // sum dy (z - mean) rstd
}

// training-mode BatchNorm: coefficients of the batch-statistics terms of dz (header: mv_bn_train_dz_coef_f32)
__global__ void bn_train_dz_coef_kernel(const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ s0,
                                        const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ scale,
                                        const float* __restrict__ count, float rows, float a, float eps, float* __restrict__ A,
                                        float* __restrict__ B, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float n = count ? count[0] : rows;
    const float k = a / n;
    const float r2 = 1.0f / (var[c] + eps);                                  // rstd^2
    const float b = -k * scale[c] * r2 * (s2[c] - mean[c] * s1[c]);
    B[c] = b;
    A[c] = -k * scale[c] * s1[c] - b * (s0[c] / n);
}

// one wave per row: dx = rstd (g dy - mean(g dy) - xhat mean(g dy xhat)); dyxhat = dy xhat (its column sums are dgamma)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ dy, float* __restrict__ dx,
                                                             float* __restrict__ dyxhat, long long M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * C;
    const float* gr = dy + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mean; q += d * d; }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
    float a = 0.f, b = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] - mean) * rstd, gd = (gamma ? gamma[c] : 1.f) * gr[c];
        a += gd; b += gd * xh;
    }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    a /= (float)C; b /= (float)C;
    for (int c = lane; c < C; c += 64) {
        const float xh = (xr[c] - mean) * rstd, gd = (gamma ? gamma[c] : 1.f) * gr[c];
        dx[row * C + c] = rstd * (gd - a - xh * b);
        dyxhat[row * C + c] = gr[c] * xh;
    }
}

// ds = scale * p * (dp - sum_j dp p)  (the softmax of scale * s, one wave per row)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                           float* __restrict__ ds, long long rows, int cols, float scale) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += p[row * cols + c] * dp[row * cols + c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    for (int c = lane; c < cols; c += 64) ds[row * cols + c] = scale * p[row * cols + c] * (dp[row * cols + c] - s);
}

// Attention backward (vit.py:64-73: softmax(q k^T scale) v with the probabilities kept) in two launches.
//   query side: one wave per (image, head, query i): dP_ij = dO_i . V_j, dS_ij = scale P_ij (dP_ij - sum_j P_ij dP_ij) -> ds[b,h,i,:]
//               (kept for the key side) and dQ_i = sum_j dS_ij K_j
//   key side:   one wave per (image, head, key j): dK_j = sum_i dS_ij Q_i, dV_j = sum_i P_ij dO_i
// qkv / dqkv rows are [q | k | v][head][dh] (3 D floats per token), dO rows D floats.  Every sum runs in index order: reproducible.
__global__ __launch_bounds__(64) void mha_bwd_q_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                       const float* __restrict__ dout, float* __restrict__ ds, float* __restrict__ dqkv,
                                                       int N, int H, int dh, float scale) {
    extern __shared__ float sm[];                          // [dh] dO_i, then [N] dS_i
    float* go = sm;
    float* dsr = sm + dh;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int D = H * dh;
    const long long tok0 = (long long)b * N;
    for (int d = lane; d < dh; d += 64) go[d] = dout[(tok0 + i) * D + h * dh + d];
    __syncthreads();
    const float* prow = probs + (((long long)b * H + h) * N + i) * N;
    float t = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float* v = qkv + (tok0 + j) * 3 * D + 2 * D + h * dh;
        float dp = 0.f;
        for (int d = 0; d < dh; ++d) dp = fmaf(go[d], v[d], dp);
        dsr[j] = dp;
        t = fmaf(prow[j], dp, t);
    }
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    float* dsg = ds + (((long long)b * H + h) * N + i) * N;
    for (int j = lane; j < N; j += 64) {
        const float v = scale * prow[j] * (dsr[j] - t);
        dsr[j] = v;
        dsg[j] = v;
    }
    __syncthreads();
    for (int d = lane; d < dh; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < N; ++j) acc = fmaf(dsr[j], qkv[(tok0 + j) * 3 * D + D + h * dh + d], acc);
        dqkv[(tok0 + i) * 3 * D + h * dh + d] = acc;
    }
}
__global__ __launch_bounds__(64) void mha_bwd_kv_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                        const float* __restrict__ dout, const float* __restrict__ ds,
                                                        float* __restrict__ dqkv, int N, int H, int dh) {
    const int j = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int D = H * dh;
    const long long tok0 = (long long)b * N;
    const float* pcol = probs + ((long long)b * H + h) * N * N + j;
    const float* dcol = ds + ((long long)b * H + h) * N * N + j;
    for (int d = lane; d < dh; d += 64) {
        float dk = 0.f, dv = 0.f;
        for (int i = 0; i < N; ++i) {
            dk = fmaf(dcol[(long long)i * N], qkv[(tok0 + i) * 3 * D + h * dh + d], dk);
            dv = fmaf(pcol[(long long)i * N], dout[(tok0 + i) * D + h * dh + d], dv);
        }
        dqkv[(tok0 + j) * 3 * D + D + h * dh + d] = dk;
        dqkv[(tok0 + j) * 3 * D + 2 * D + h * dh + d] = dv;
    }
}

// optax.softmax_cross_entropy(logits, onehot).mean(): loss = mean_b (logsumexp(l_b) - sum_k t_bk l_bk); dlogits = (softmax - t) / B
__global__ __launch_bounds__(64) void softmax_xent_kernel(const float* __restrict__ logits, const float* __restrict__ target,
                                                           float* __restrict__ loss_rows, float* __restrict__ dlogits, int B, int K) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* l = logits + (long long)b * K;
    const float* t = target + (long long)b * K;
    float mx = -3.0e38f;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, l[k]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float se = 0.f, tl = 0.f, ts = 0.f;
    for (int k = lane; k < K; k += 64) { se += expf(l[k] - mx); tl += t[k] * l[k]; ts += t[k]; }
    for (int o = 32; o > 0; o >>= 1) { se += __shfl_xor(se, o); tl += __shfl_xor(tl, o); ts += __shfl_xor(ts, o); }
    const float lse = mx + logf(se);
    if (lane == 0) loss_rows[b] = ts * lse - tl;                       // -sum_k t_k log_softmax_k
    const float invB = 1.0f / (float)B;
    for (int k = lane; k < K; k += 64) dlogits[(long long)b * K + k] = (ts * expf(l[k] - lse) - t[k]) * invB;
}

__global__ void mean_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += v[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

// optax.adam: m = b1 m + (1 - b1) g; v = b2 v + (1 - b2) g^2; update = -lr (m / bc1) / (sqrt(v / bc2) + eps)
__global__ void adam_kernel(const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, float* __restrict__ upd,
                            long long n, float lr, float b1, float b2, float eps, float bc1, float bc2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        upd[i] = -lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    }
}


// Swin shifted-window attention backward (swin.py:123-250), one workgroup per (window, head, image), one thread per token.
// qkv NHWC [B,Hf,Wf,3C] ([q|k|v][head][dh]); roll / partition / mask are the forward's index arithmetic (generic.hip:
// swin_attn_generic_kernel).  With s = (q / sqrt(dh)) . k + bias + mask, p = softmax(s), o = p v:
//   g = p * (dp - sum_j dp p), dp = do . v^T;  dq = g k / sqrt(dh);  dk = g^T (q / sqrt(dh));  dv = p^T do;  dbias += g.
// g is also written out per window ([B * windows][heads][n * n]) so that the bias gradient is a deterministic column sum.
__global__ __launch_bounds__(64) void swin_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ bias,
                                                            const float* __restrict__ dout, float* __restrict__ dqkv,
                                                            float* __restrict__ gall, int B, int Hf, int Wf, int C, int heads, int wsh,
                                                            int wsw, int shh, int shw) {
    extern __shared__ float sm[];
    const int n = wsh * wsw, dh = C / heads;
    float* q = sm;                  // [n][dh] (scaled)
    float* k = q + n * dh;
    float* v = k + n * dh;
    float* go = v + n * dh;         // dout rows
    float* P = go + n * dh;         // [n][n]
    float* G = P + n * n;
    const int nWw = Wf / wsw;
    const int win = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int wy = win / nWw, wx = win % nWw;
    const int t = threadIdx.x;
    const float scale = rsqrtf((float)dh);
    const bool shifted = (shh + shw) > 0;
    auto region = [&](int y, int x) {
        const int rh = (y < Hf - wsh) ? 0 : (y < Hf - shh ? 1 : 2);
        const int rw = (x < Wf - wsw) ? 0 : (x < Wf - shw ? 1 : 2);
        return rh * 3 + rw;
    };
    long long pos = 0;
    int myreg = 0;
    if (t < n) {
        const int ty = wy * wsh + t / wsw, tx = wx * wsw + t % wsw;          // rolled coordinates
        myreg = region(ty, tx);
        pos = ((long long)b * Hf + (ty + shh) % Hf) * Wf + (tx + shw) % Wf;
        const float* r = qkv + pos * 3 * C + h * dh;
        for (int d = 0; d < dh; ++d) {
            q[t * dh + d] = r[d] * scale;
            k[t * dh + d] = r[C + d];
            v[t * dh + d] = r[2 * C + d];
            go[t * dh + d] = dout[pos * C + h * dh + d];
        }
    }
    __shared__ int reg[64];
    if (t < 64) reg[t] = myreg;
    __syncthreads();
    if (t < n) {
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) {
            float s = 0.f;
            for (int d = 0; d < dh; ++d) s = fmaf(q[t * dh + d], k[j * dh + d], s);
            s += bias[((long long)h * n + t) * n + j];
            if (shifted && reg[j] != myreg) s += -100.0f;
            P[t * n + j] = s;
            mx = fmaxf(mx, s);
        }
        float sum = 0.f;
        for (int j = 0; j < n; ++j) { const float e = __expf(P[t * n + j] - mx); P[t * n + j] = e; sum += e; }
        const float inv = 1.f / sum;
        float rs = 0.f;
        for (int j = 0; j < n; ++j) {
            const float pj = P[t * n + j] * inv;
            float dp = 0.f;
            for (int d = 0; d < dh; ++d) dp = fmaf(go[t * dh + d], v[j * dh + d], dp);
            P[t * n + j] = pj;
            G[t * n + j] = dp;
            rs = fmaf(dp, pj, rs);
        }
        float* gw = gall + (((long long)b * gridDim.x + win) * heads + h) * n * n + (long long)t * n;
        for (int j = 0; j < n; ++j) {
            const float g = P[t * n + j] * (G[t * n + j] - rs);
            G[t * n + j] = g;
            gw[j] = g;
        }
        float* dq = dqkv + pos * 3 * C + h * dh;
        for (int d = 0; d < dh; ++d) {
            float a = 0.f;
            for (int j = 0; j < n; ++j) a = fmaf(G[t * n + j], k[j * dh + d], a);
            dq[d] = a * scale;
        }
    }
    __syncthreads();
    if (t < n) {
        float* dk = dqkv + pos * 3 * C + C + h * dh;
        float* dv = dqkv + pos * 3 * C + 2 * C + h * dh;
        for (int d = 0; d < dh; ++d) {
            float a = 0.f, c = 0.f;
            for (int i = 0; i < n; ++i) {
                a = fmaf(G[i * n + t], q[i * dh + d], a);          // q is already scaled
                c = fmaf(P[i * n + t], go[i * dh + d], c);
            }
            dk[d] = a;
            dv[d] = c;
        }
    }
}

// out[t][c] = sum over the rows m with idx[m] == t of src[m][c]  (relative_position_bias_table gradient: rows = the n * n
// (query, key) pairs, columns = heads); one block per table row, fixed summation order
__global__ void scatter_rows_sum_kernel(const float* __restrict__ src, const int* __restrict__ idx, float* __restrict__ out, int M,
                                        int Ccols) {
    const int t = blockIdx.x;
    for (int c = threadIdx.x; c < Ccols; c += blockDim.x) {
        float a = 0.f;
        for (int m = 0; m < M; ++m)
            if (idx[m] == t) a += src[(long long)m * Ccols + c];
        out[(long long)t * Ccols + c] = a;
    }
}

// inverse of the 2 x 2 neighbourhood gather of Swin's patch merging (swin.py:23-31): dx[b, 2 ho + (q & 1), 2 wo + (q >> 1), c] =
// dy[b, ho, wo, q C + c]
__global__ void patch_merge_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const long long n = (long long)B * H * W * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long m = i / C;
        const int wi = (int)(m % W);
        m /= W;
        const int hi = (int)(m % H);
        const int b = (int)(m / H);
        const int q = (hi & 1) + 2 * (wi & 1);
        dx[i] = dy[(((long long)b * Ho + (hi >> 1)) * Wo + (wi >> 1)) * 4 * C + q * C + c];
    }
}

__global__ void transpose2d_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C, long long xs) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < R && c < C) ? x[(long long)r * xs + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (c < C && r < R) y[(long long)c * R + r] = tile[threadIdx.x][j];
    }
}

inline unsigned blocks_for(long long n, int per) {
    long long b = (n + per - 1) / per;
    return (unsigned)(b < 1 ? 1 : (b > 65535 * 16 ? 65535 * 16 : b));
}

}  // namespace

}  // namespace mv

extern "C" {

using namespace mv;

int mv_conv2d_dgrad_nhwc_f32(const float* dy, const float* w_krsc, float* dx, int N, int H, int W, int C, int K, int R, int S, int sh,
                             int sw, int ph, int pw, int dh, int dw, int groups, mv_stream_t stream) {
    MV_CHECK_ARG(dy && w_krsc && dx && N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0,
                 "conv2d_dgrad: bad arguments");
    MV_CHECK_ARG(groups > 0 && C % groups == 0 && K % groups == 0, "conv2d_dgrad: groups = %d does not divide C = %d, K = %d", groups, C, K);
    const int Ho = (H + 2 * ph - dh * (R - 1) - 1) / sh + 1, Wo = (W + 2 * pw - dw * (S - 1) - 1) / sw + 1;
    MV_CHECK_ARG(Ho > 0 && Wo > 0 && (long long)N * H * W < (1LL << 31), "conv2d_dgrad: bad dims");
    if (groups == 1 && C >= 8 && !get_flag("dgrad_valu")) {
        const long long M = (long long)N * H * W;
        set_kernel_name("conv_dgrad_mfma_f32");
        hipLaunchKernelGGL(conv_dgrad_mfma_kernel, dim3((unsigned)((M + 127) / 128), (unsigned)((C + 31) / 32)), dim3(256), 0,
                           (hipStream_t)stream, dy, w_krsc, dx, N, H, W, C, K, R, S, Ho, Wo, sh, sw, ph, pw, dh, dw);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    set_kernel_name("conv_dgrad_f32");
    hipLaunchKernelGGL(conv_dgrad_kernel, dim3((unsigned)((long long)N * H * W)), dim3(C >= 256 ? 256 : (C > 64 ? 128 : 64)), 0,
                       (hipStream_t)stream, dy, w_krsc, dx, N, H, W, C, K, R, S, Ho, Wo, sh, sw, ph, pw, dh, dw, groups);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_conv2d_wgrad_nhwc_f32(const float* x, const float* dy, float* dw_krsc, int N, int H, int W, int C, int K, int R, int S, int sh,
                             int sw, int ph, int pw, int dh, int dw, int groups, mv_stream_t stream) {
    MV_CHECK_ARG(x && dy && dw_krsc && N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && R > 0 && S > 0 && sh > 0 && sw > 0 && dh > 0 && dw > 0,
                 "conv2d_wgrad: bad arguments");
    MV_CHECK_ARG(groups > 0 && C % groups == 0 && K % groups == 0, "conv2d_wgrad: groups = %d does not divide C = %d, K = %d", groups, C, K);
    const int Ho = (H + 2 * ph - dh * (R - 1) - 1) / sh + 1, Wo = (W + 2 * pw - dw * (S - 1) - 1) / sw + 1;
    MV_CHECK_ARG(Ho > 0 && Wo > 0 && (long long)K * R * S < (1LL << 31), "conv2d_wgrad: bad dims");
    const int cg = C / groups, kg = K / groups;
    if (!get_flag("wgrad_valu")) {                  // (the one-thread-per-weight kernel below stays as the on-device cross-check)
        // matrix-core path: tiles x position chunks ~ 4096 waves; the chunks' partial sums live in the caller's scratch (mv_set_scratch),
        // without scratch one chunk per tile writes the result directly
        const long long P = (long long)N * Ho * Wo, out_elems = (long long)K * R * S * cg;
        const int ktiles = (kg + 31) / 32, ctiles = (cg + 31) / 32;
        const long long tiles = (long long)groups * ktiles * ctiles * R * S;
        long long split = tiles >= 2048 ? 1 : (4096 + tiles - 1) / tiles;
        if (split > (P + 127) / 128) split = (P + 127) / 128;
        if (split > 1024) split = 1024;
        float* part = dw_krsc;
        if (split > 1) {
            // partial sums live BEHIND the first 4096 bytes of the offer: those are the arrival words of the forward's split-K
            // protocol (header: mv_set_scratch) and stay zero, so an offer this entry leaves unused can never hand a later
            // split-K launch a dirty sync area
            size_t have = 0;
            void* sc = peek_scratch((hipStream_t)stream, &have);
            have = have > SCRATCH_SYNC_BYTES ? have - SCRATCH_SYNC_BYTES : 0;
            const long long fit = sc ? (long long)(have / ((size_t)out_elems * 4)) : 0;
            if (fit < 2) split = 1;
            else {
                if (split > fit) split = fit;
                part = (float*)((char*)take_scratch((hipStream_t)stream, SCRATCH_SYNC_BYTES + (size_t)split * out_elems * 4) + SCRATCH_SYNC_BYTES);
            }
        }
        const long long chunk = ((P + split - 1) / split + 1) & ~1LL;                 // even: a pair of positions never straddles two chunks
        set_kernel_name("conv_wgrad_mfma_f32");
        hipLaunchKernelGGL(conv_wgrad_mfma_kernel, dim3((unsigned)tiles, (unsigned)split), dim3(64), 0, (hipStream_t)stream, x, dy, part, N, H,
                           W, C, K, R, S, Ho, Wo, sh, sw, ph, pw, dh, dw, groups, P, chunk, ktiles, ctiles);
        if (split > 1)
            hipLaunchKernelGGL(sum_parts_kernel, dim3(blocks_for(out_elems, 256)), dim3(256), 0, (hipStream_t)stream, part, dw_krsc,
                               out_elems, (int)split);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    set_kernel_name("conv_wgrad_f32");
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3((unsigned)(K * R * S)), dim3(cg >= 256 ? 256 : (cg > 64 ? 128 : 64)), 0,
                       (hipStream_t)stream, x, dy, dw_krsc, N, H, W, C, K, R, S, Ho, Wo, sh, sw, ph, pw, dh, dw, groups);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_act_bwd_f32(const float* dy, const float* ref, float* dx, int64_t n, int act, mv_stream_t stream) {
    MV_CHECK_ARG(dy && ref && dx && n > 0, "act_bwd: bad arguments");
    MV_CHECK_ARG(act >= MV_ACT_NONE && act <= MV_ACT_SILU, "act_bwd: unknown activation %d", act);
    set_kernel_name("act_bwd_f32");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, ref, dx, (long long)n, act);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_maxpool2d_bwd_nhwc_f32(const float* x, const float* dy, float* dx, int N, int H, int W, int C, int kh, int kw, int sh, int sw,
                              int ph, int pw, mv_stream_t stream) {
    MV_CHECK_ARG(x && dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && kh > 0 && kw > 0 && sh > 0 && sw > 0, "maxpool2d_bwd: bad arguments");
    const int Ho = (H + 2 * ph - kh) / sh + 1, Wo = (W + 2 * pw - kw) / sw + 1;
    MV_CHECK_ARG(Ho > 0 && Wo > 0 && (long long)N * H * W < (1LL << 31), "maxpool2d_bwd: bad dims");
    set_kernel_name("maxpool_bwd_f32");
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)((long long)N * H * W)), dim3(C >= 256 ? 256 : 64), 0, (hipStream_t)stream, x,
                       dy, dx, N, H, W, C, kh, kw, sh, sw, ph, pw, Ho, Wo);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_avgpool_global_bwd_nhwc_f32(const float* dy, float* dx, int N, int HW, int C, mv_stream_t stream) {
    MV_CHECK_ARG(dy && dx && N > 0 && HW > 0 && C > 0, "avgpool_global_bwd: bad arguments");
    const long long n = (long long)N * HW * C;
    set_kernel_name("avgpool_global_bwd_f32");
    hipLaunchKernelGGL(avgpool_global_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, HW, C, n);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_colsum_f32(const float* a, const float* b, float* out, int64_t M, int C, mv_stream_t stream) {
    MV_CHECK_ARG(a && out && M > 0 && C > 0, "colsum: bad arguments");
    // many rows, few columns (bias / BatchNorm gradients over a feature map): rows split over blocks, parts in the caller's scratch
    long long split = M >= 512 ? (M + 255) / 256 : 1;
    if (split > 512) split = 512;
    if (split > 1) {
        char* sc = (char*)take_scratch((hipStream_t)stream, SCRATCH_SYNC_BYTES + (size_t)split * C * 4);
        float* part = sc ? (float*)(sc + SCRATCH_SYNC_BYTES) : nullptr;          // behind the split-K sync area (see conv wgrad)
        if (part) {
            const long long chunk = (M + split - 1) / split;
            set_kernel_name("colsum_split_f32");
            hipLaunchKernelGGL(colsum_part_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)split), dim3(256), 0, (hipStream_t)stream, a, b,
                               part, (long long)M, C, chunk);
            hipLaunchKernelGGL(sum_parts_kernel, dim3(blocks_for(C, 256)), dim3(256), 0, (hipStream_t)stream, part, out, (long long)C,
                               (int)split);
            MV_LAUNCH_CHECK();
            return MV_OK;
        }
    }
    set_kernel_name("colsum_f32");
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long long)M, C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_channel_scale_bwd_f32(const float* g, const float* x, float* ds, int B, int HW, int C, mv_stream_t stream) {
    MV_CHECK_ARG(g && x && ds && B > 0 && B <= 65535 && HW > 0 && C > 0, "channel_scale_bwd: bad arguments");
    set_kernel_name("channel_scale_bwd_f32");
    hipLaunchKernelGGL(channel_scale_bwd_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)B), dim3(256), 0, (hipStream_t)stream, g, x, ds,
                       HW, C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_bn_dgamma_f32(const float* sum_dy_z, const float* sum_dy, const float* mean, const float* var, float eps, float* dgamma, int C,
                     mv_stream_t stream) {
    MV_CHECK_ARG(sum_dy_z && sum_dy && mean && var && dgamma && C > 0, "bn_dgamma: bad arguments");
    set_kernel_name("bn_dgamma_f32");
    hipLaunchKernelGGL(bn_dgamma_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sum_dy_z, sum_dy, mean,
                       var, eps, dgamma, C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_bn_train_dz_coef_f32(const float* sum_dy, const float* sum_dy_z, const float* sum_z, const float* mean, const float* var,
                            const float* scale, const float* count, float rows, float a, float eps, float* A, float* B, int C,
                            mv_stream_t stream) {
    MV_CHECK_ARG(sum_dy && sum_dy_z && sum_z && mean && var && scale && A && B && C > 0 && (count || rows > 0.f),
                 "bn_train_dz_coef: bad arguments");
    set_kernel_name("bn_train_dz_coef_f32");
    hipLaunchKernelGGL(bn_train_dz_coef_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sum_dy, sum_dy_z,
                       sum_z, mean, var, scale, count, rows, a, eps, A, B, C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_layernorm_bwd_f32(const float* x, const float* gamma, const float* dy, float* dx, float* dy_xhat, int64_t M, int C, float eps,
                         mv_stream_t stream) {
    MV_CHECK_ARG(x && dy && dx && dy_xhat && M > 0 && C > 0, "layernorm_bwd: bad arguments");
    set_kernel_name("layernorm_bwd_f32");
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, dy, dx, dy_xhat,
                       (long long)M, C, eps);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_softmax_bwd_f32(const float* p, const float* dp, float* ds, int64_t rows, int cols, float scale, mv_stream_t stream) {
    MV_CHECK_ARG(p && dp && ds && rows > 0 && cols > 0, "softmax_bwd: bad arguments");
    set_kernel_name("softmax_bwd_f32");
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, dp, ds,
                       (long long)rows, cols, scale);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_mha_bwd_f32(const float* qkv, const float* probs, const float* dout, float* ds_scratch, float* dqkv, int B, int N, int H, int dh,
                   float scale, mv_stream_t stream) {
    MV_CHECK_ARG(qkv && probs && dout && ds_scratch && dqkv && B > 0 && N > 0 && H > 0 && dh > 0 && B <= 65535 && H <= 65535,
                 "mha_bwd: bad arguments");
    MV_CHECK_ARG((size_t)(dh + N) * 4 <= 64 * 1024, "mha_bwd: N = %d tokens do not fit the row buffer", N);
    set_kernel_name("mha_bwd_f32");
    const dim3 grid((unsigned)N, (unsigned)H, (unsigned)B);
    hipLaunchKernelGGL(mha_bwd_q_kernel, grid, dim3(64), (size_t)(dh + N) * 4, (hipStream_t)stream, qkv, probs, dout, ds_scratch, dqkv, N, H,
                       dh, scale);
    hipLaunchKernelGGL(mha_bwd_kv_kernel, grid, dim3(64), 0, (hipStream_t)stream, qkv, probs, dout, ds_scratch, dqkv, N, H, dh);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_softmax_xent_f32(const float* logits, const float* target, float* loss_rows, float* loss_mean, float* dlogits, int B, int K,
                        mv_stream_t stream) {
    MV_CHECK_ARG(logits && target && loss_rows && loss_mean && dlogits && B > 0 && K > 0, "softmax_xent: bad arguments");
    set_kernel_name("softmax_xent_f32");
    hipLaunchKernelGGL(softmax_xent_kernel, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, logits, target, loss_rows, dlogits, B, K);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, loss_rows, loss_mean, B);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_adam_step_f32(const float* grad, float* m, float* v, float* update, int64_t n, float lr, float b1, float b2, float eps,
                     float bias_corr1, float bias_corr2, mv_stream_t stream) {
    MV_CHECK_ARG(grad && m && v && update && n > 0 && bias_corr1 > 0.f && bias_corr2 > 0.f, "adam_step: bad arguments");
    set_kernel_name("adam_step_f32");
    hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, grad, m, v, update, (long long)n, lr, b1,
                       b2, eps, bias_corr1, bias_corr2);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_swin_window_attn_bwd_f32(const float* qkv, const float* bias, const float* dout, float* dqkv, float* g_windows, int B, int Hf,
                                int Wf, int C, int heads, int ws_h, int ws_w, int shift_h, int shift_w, mv_stream_t stream) {
    MV_CHECK_ARG(qkv && bias && dout && dqkv && g_windows && B > 0 && Hf > 0 && Wf > 0 && C > 0 && heads > 0, "swin_attn_bwd: bad args");
    MV_CHECK_ARG(C % heads == 0 && ws_h > 0 && ws_w > 0 && Hf % ws_h == 0 && Wf % ws_w == 0 && ws_h * ws_w <= 64,
                 "swin_attn_bwd: map %dx%d / window %dx%d (<= 64 tokens) / %d heads", Hf, Wf, ws_h, ws_w, heads);
    MV_CHECK_ARG(shift_h >= 0 && shift_w >= 0 && shift_h < ws_h && shift_w < ws_w && heads <= 65535 && B <= 65535, "swin_attn_bwd: bad shift / grid");
    if (ws_h >= Hf) shift_h = 0;  // swin.py:116-120
    if (ws_w >= Wf) shift_w = 0;
    const int n = ws_h * ws_w, dh = C / heads;
    const size_t smem = (size_t)(4 * n * dh + 2 * n * n) * sizeof(float);
    MV_CHECK_ARG(smem <= 150 * 1024, "swin_attn_bwd: head width %d too large", dh);
    auto kern = swin_attn_bwd_kernel;
    MV_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    set_kernel_name("swin_attn_bwd_f32");
    hipLaunchKernelGGL(kern, dim3((unsigned)((Hf / ws_h) * (Wf / ws_w)), (unsigned)heads, (unsigned)B), dim3(64), smem, (hipStream_t)stream,
                       qkv, bias, dout, dqkv, g_windows, B, Hf, Wf, C, heads, ws_h, ws_w, shift_h, shift_w);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_scatter_rows_sum_f32(const float* src, const int* index, float* out, int M, int C, int T, mv_stream_t stream) {
    MV_CHECK_ARG(src && index && out && M > 0 && C > 0 && T > 0, "scatter_rows_sum: bad arguments");
    set_kernel_name("scatter_rows_sum_f32");
    hipLaunchKernelGGL(scatter_rows_sum_kernel, dim3((unsigned)T), dim3(64), 0, (hipStream_t)stream, src, index, out, M, C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_patch_merge_gather_bwd_f32(const float* dy, float* dx, int B, int H, int W, int C, mv_stream_t stream) {
    MV_CHECK_ARG(dy && dx && B > 0 && H > 0 && W > 0 && C > 0, "patch_merge_bwd: bad arguments");
    const long long n = (long long)B * H * W * C;
    set_kernel_name("patch_merge_bwd_f32");
    hipLaunchKernelGGL(patch_merge_bwd_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, B, H, W, C);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_transpose2d_f32(const float* x, float* y, int R, int C, int64_t x_row_stride, mv_stream_t stream) {
    MV_CHECK_ARG(x && y && R > 0 && C > 0, "transpose2d: bad arguments");
    const long long xs = x_row_stride ? (long long)x_row_stride : (long long)C;
    MV_CHECK_ARG(xs >= C && (R + 31) / 32 <= 65535, "transpose2d: bad dims");
    set_kernel_name("transpose2d_f32");
    hipLaunchKernelGGL(transpose2d_kernel, dim3((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32)), dim3(32, 8), 0, (hipStream_t)stream,
                       x, y, R, C, xs);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

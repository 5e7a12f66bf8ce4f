// Training-mode Dropout (eqx.nn.Dropout with inference = False, as the reference's classifiers / transformer blocks hold it:
// alexnet.py:63-68, vit.py:39-53, mlps.py:43-52): y = where(bernoulli(key, 1 - p, x.shape), x / (1 - p), 0), one key per sample
// (the caller vmaps over the keys).  The mask is JAX's bit stream (jax.random.bernoulli = uniform(key, shape) < q; uniform =
// mantissa bits of Threefry-2x32 words in the counter layout of `_threefry_random_bits`, see eqxvision_amd/random.py), generated
// here per element: element i of the LOGICAL single-sample array (row-major over the reference's (C,H,W) / (N,D) / (D,) shape)
// is word i of the sample's stream.  Read x, write y; the ~110 integer operations of a Threefry call per element (55 when a call
// serves two elements, dropout_pair_kernel) are what bounds it: 2.5 TB/s of the 4-5 a pure stream reaches.
#include "mfma_common.h"
#include "rng_common.h"

namespace mv {

template <typename T>
__global__ void dropout_kernel(const T* x, const uint32_t* keys, T* y, long long per, int C, long long HW, int chw, float q,
                               long long total) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const long long b = g / per, pi = g - b * per;      // physical index inside the sample: pixel-major, channel fastest
        long long li = pi;
        if (chw) {                                          // logical (C, H, W) order
            const long long hw = pi / C;
            const int c = (int)(pi - hw * C);
            li = (long long)c * HW + hw;
        }
        const uint32_t w = stream_word(keys[2 * b], keys[2 * b + 1], (uint32_t)li, (uint32_t)per);
        const float u = __uint_as_float((w >> 9) | 0x3F800000u) - 1.0f;
        float v;
        if constexpr (sizeof(T) == 2) v = __uint_as_float((uint32_t)x[g] << 16);
        else v = x[g];
        v = u < q ? v / q : 0.f;
        if constexpr (sizeof(T) == 2) y[g] = (T)(pack_bf2(v, 0.f) & 0xffffu);
        else y[g] = v;
    }
}

// 8 physically consecutive values per thread (16-byte accesses for bf16, 2 x 16 for fp32): C % 8 == 0, so in the NHWC / (C,H,W)
// case they are 8 channels of one pixel -- logical indices HW apart
template <typename T>
__global__ void dropout_vec8_kernel(const T* x, const uint32_t* keys, T* y, long long per, int C, long long HW, int chw, float q,
                                    long long total8) {
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total8; g += (long long)gridDim.x * blockDim.x) {
        const long long e0 = g * 8;
        const long long b = e0 / per, pi = e0 - b * per;
        const uint32_t k0 = keys[2 * b], k1 = keys[2 * b + 1];
        long long li = pi, lstep = 1;
        if (chw) {
            const long long hw = pi / C;
            li = (pi - hw * C) * HW + hw;
            lstep = HW;
        }
        float v[8];
        if constexpr (sizeof(T) == 2) {
            const uint4 u = *(const uint4*)(x + e0);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] = __uint_as_float(w[e] << 16);
                v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u);
            }
        } else {
            const float4 a = *(const float4*)(x + e0), c = *(const float4*)(x + e0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t w = stream_word(k0, k1, (uint32_t)(li + e * lstep), (uint32_t)per);
            const float u = __uint_as_float((w >> 9) | 0x3F800000u) - 1.0f;
            v[e] = u < q ? v[e] / q : 0.f;
        }
        Out8<T>::st(y + e0, v);
    }
}

// One Threefry call yields TWO words of the stream: those of logical elements i and i + n/2.  When n is even and the two
// elements' 8-value vectors are both 16-byte aligned -- physically n/2 apart (row-major samples) or C/2 channels apart at the same
// pixel ((C,H,W) samples stored NHWC: (c + C/2) * HW + hw = i + n/2) -- a thread produces both vectors from 8 calls: half the
// integer work, which is what bounds this kernel (the plain x8 kernel runs at 1.5 TB/s).
template <typename T>
__global__ void dropout_pair_kernel(const T* x, const uint32_t* keys, T* y, long long per, int C, long long HW, int chw, float q,
                                    long long pairs_per_sample, long long total_pairs) {
    const uint32_t half = (uint32_t)(per >> 1);
    const long long poff = chw ? (long long)(C >> 1) : (long long)half;      // physical distance of the partner vector
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total_pairs; g += (long long)gridDim.x * blockDim.x) {
        const long long b = g / pairs_per_sample, j = g - b * pairs_per_sample;
        long long pi, li, lstep = 1;
        if (chw) {
            const long long vpp = C >> 4;                                    // first-half vectors per pixel
            const long long hw = j / vpp;
            const long long c0 = (j - hw * vpp) * 8;
            pi = hw * C + c0;
            li = c0 * HW + hw;
            lstep = HW;
        } else {
            pi = j * 8;
            li = pi;
        }
        const uint32_t k0 = keys[2 * b], k1 = keys[2 * b + 1];
        const T* xa = x + b * per + pi;
        T* ya = y + b * per + pi;
        float va[8], vb[8];
        if constexpr (sizeof(T) == 2) {
            const uint4 ua = *(const uint4*)xa, ub = *(const uint4*)(xa + poff);
            const uint32_t wa[4] = {ua.x, ua.y, ua.z, ua.w}, wb[4] = {ub.x, ub.y, ub.z, ub.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                va[2 * e] = __uint_as_float(wa[e] << 16); va[2 * e + 1] = __uint_as_float(wa[e] & 0xffff0000u);
                vb[2 * e] = __uint_as_float(wb[e] << 16); vb[2 * e + 1] = __uint_as_float(wb[e] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 a = *(const float4*)(xa + 4 * h), c = *(const float4*)(xa + poff + 4 * h);
                va[4 * h] = a.x; va[4 * h + 1] = a.y; va[4 * h + 2] = a.z; va[4 * h + 3] = a.w;
                vb[4 * h] = c.x; vb[4 * h + 1] = c.y; vb[4 * h + 2] = c.z; vb[4 * h + 3] = c.w;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            uint32_t x0 = (uint32_t)(li + e * lstep), x1 = x0 + half;       // both < n: n is even, no padding counter
            threefry2x32(k0, k1, x0, x1);
            const float u0 = __uint_as_float((x0 >> 9) | 0x3F800000u) - 1.0f;
            const float u1 = __uint_as_float((x1 >> 9) | 0x3F800000u) - 1.0f;
            va[e] = u0 < q ? va[e] / q : 0.f;
            vb[e] = u1 < q ? vb[e] / q : 0.f;
        }
        Out8<T>::st(ya, va);
        Out8<T>::st(ya + poff, vb);
    }
}

}  // namespace mv

using namespace mv;

// Dropout of an NHWC map whose LOGICAL layout is the reference's window partition of the cyclically shifted map
// ((num_windows, tokens, C), swin.py:150-160): `_func_dropout(x, dropout, key)` after the attention's projection (swin.py:233),
// before the windows are put back.  Physical pixel (oy, ox) sits at rolled position p = (o - shift) mod size, i.e. in window
// (py / wsh, px / wsw) as token (py % wsh) * wsw + px % wsw.  8 channels per thread.
template <typename T>
__global__ void dropout_windows_kernel(const T* x, const uint32_t* keys, T* y, int Hf, int Wf, int C, int wsh, int wsw, int shh,
                                       int shw, float q, long long total8) {
    const long long per = (long long)Hf * Wf * C;
    const int n = wsh * wsw, nWw = Wf / wsw;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total8; g += (long long)gridDim.x * blockDim.x) {
        const long long e0 = g * 8, pix = e0 / C;
        const int c = (int)(e0 - pix * C);
        const long long b = pix / ((long long)Hf * Wf);
        const int r = (int)(pix - b * Hf * Wf), oy = r / Wf, ox = r - oy * Wf;
        const int py = oy - shh + (oy < shh ? Hf : 0), px = ox - shw + (ox < shw ? Wf : 0);
        const int w = (py / wsh) * nWw + px / wsw, t = (py % wsh) * wsw + px % wsw;
        const uint32_t li = (uint32_t)(((long long)w * n + t) * C + c);
        const uint32_t k0 = keys[2 * b], k1 = keys[2 * b + 1];
        float v[8];
        if constexpr (sizeof(T) == 2) {
            const uint4 u = *(const uint4*)(x + e0);
            const uint32_t wd[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] = __uint_as_float(wd[e] << 16);
                v[2 * e + 1] = __uint_as_float(wd[e] & 0xffff0000u);
            }
        } else {
            const float4 a = *(const float4*)(x + e0), d = *(const float4*)(x + e0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            v[e] = word_uniform01(stream_word(k0, k1, li + e, (uint32_t)per)) < q ? v[e] / q : 0.f;
        Out8<T>::st(y + e0, v);
    }
}

// DropPath's noise (drop_path.py:51-61) as the [B][C] scale the broadcast-multiply kernel takes: bernoulli(key_b, keep) for the whole
// sample ("global": the single word of a 1-word stream, repeated over C) or bernoulli(key_b, keep, (C,)) per entry of the first
// axis ("local": word c of a C-word stream), divided by keep
template <typename T>
__global__ void drop_path_noise_kernel(const uint32_t* keys, T* out, int B, int C, int local, float keep) {
    const long long total = (long long)B * C;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(g / C), c = (int)(g - (long long)b * C);
        const uint32_t w = local ? stream_word(keys[2 * b], keys[2 * b + 1], (uint32_t)c, (uint32_t)C)
                                 : stream_word(keys[2 * b], keys[2 * b + 1], 0u, 1u);
        const float v = word_uniform01(w) < keep ? 1.f / keep : 0.f;
        if constexpr (sizeof(T) == 2) out[g] = (T)(pack_bf2(v, 0.f) & 0xffffu);
        else out[g] = v;
    }
}

// jax.random.split of R keys at once: child i of key r = words (2i, 2i + 1) of r's 2 * num-word stream (a thread per child)
__global__ void prng_split_kernel(const uint32_t* keys, uint32_t* out, long long R, int num, int child_major) {
    const long long total = R * num;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const long long r = g / num;
        const int i = (int)(g - r * num);
        const uint32_t k0 = keys[2 * r], k1 = keys[2 * r + 1], n = 2u * (uint32_t)num;
        const long long o = child_major ? ((long long)i * R + r) * 2 : g * 2;
        out[o] = stream_word(k0, k1, 2u * i, n);
        out[o + 1] = stream_word(k0, k1, 2u * i + 1, n);
    }
}

extern "C" {

int mv_drop_path_noise(const void* keys, void* out, int B, int C, int local, float keep_prob, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(keys && out && B > 0 && C > 0, "drop_path_noise: bad args");
    MV_CHECK_ARG(keep_prob > 0.f && keep_prob <= 1.f, "drop_path_noise: keep probability %g outside (0, 1]", keep_prob);
    MV_CHECK_ARG(dtype == MV_BF16 || dtype == MV_F32, "drop_path_noise: unknown dtype %d", dtype);
    long long g = ((long long)B * C + 255) / 256;
    const unsigned grid = (unsigned)(g > 4096 ? 4096 : g);
    set_kernel_name("drop_path_noise");
    if (dtype == MV_F32)
        hipLaunchKernelGGL(drop_path_noise_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)keys,
                           (float*)out, B, C, local, keep_prob);
    else
        hipLaunchKernelGGL(drop_path_noise_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)keys,
                           (bf16_t*)out, B, C, local, keep_prob);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_prng_split(const void* keys, void* out, int64_t R, int num, int child_major, mv_stream_t stream) {
    MV_CHECK_ARG(keys && out && keys != out, "prng_split: NULL or aliased pointers");
    MV_CHECK_ARG(R > 0 && num > 0 && num < (1 << 30), "prng_split: bad dims R=%lld num=%d", (long long)R, num);
    long long g = (R * num + 255) / 256;
    set_kernel_name("prng_split");
    hipLaunchKernelGGL(prng_split_kernel, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)keys, (uint32_t*)out, (long long)R, num, child_major);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_dropout_windows_fwd(const void* x, const void* keys, void* y, int B, int Hf, int Wf, int C, int ws_h, int ws_w, int shift_h,
                           int shift_w, float keep_prob, int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && keys && y, "dropout_windows: NULL pointer");
    MV_CHECK_ARG(B > 0 && Hf > 0 && Wf > 0 && C > 0 && C % 8 == 0, "dropout_windows: bad dims (C must be a multiple of 8)");
    MV_CHECK_ARG(ws_h > 0 && ws_w > 0 && Hf % ws_h == 0 && Wf % ws_w == 0, "dropout_windows: map %dx%d is not a multiple of the window %dx%d",
                 Hf, Wf, ws_h, ws_w);
    MV_CHECK_ARG(shift_h >= 0 && shift_w >= 0 && shift_h < ws_h && shift_w < ws_w, "dropout_windows: bad shift");
    MV_CHECK_ARG((long long)Hf * Wf * C < (1LL << 32), "dropout_windows: more than 2^32 values per sample");
    MV_CHECK_ARG(keep_prob > 0.f && keep_prob <= 1.f, "dropout_windows: keep probability %g outside (0, 1]", keep_prob);
    MV_CHECK_ARG(dtype == MV_BF16 || dtype == MV_F32, "dropout_windows: unknown dtype %d", dtype);
    if (ws_h >= Hf) shift_h = 0;  // swin.py:116-120
    if (ws_w >= Wf) shift_w = 0;
    const long long total8 = (long long)B * Hf * Wf * C / 8;
    long long gv = (total8 + 255) / 256;
    const int gridv = (int)(gv > 256 * 32 ? 256 * 32 : gv);
    set_kernel_name("dropout_threefry_windows");
    if (dtype == MV_F32)
        hipLaunchKernelGGL(dropout_windows_kernel<float>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                           (const uint32_t*)keys, (float*)y, Hf, Wf, C, ws_h, ws_w, shift_h, shift_w, keep_prob, total8);
    else
        hipLaunchKernelGGL(dropout_windows_kernel<bf16_t>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (const uint32_t*)keys, (bf16_t*)y, Hf, Wf, C, ws_h, ws_w, shift_h, shift_w, keep_prob, total8);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_dropout_fwd(const void* x, const void* keys, void* y, int B, int64_t per_sample, int C, int chw_logical, float keep_prob,
                   int dtype, mv_stream_t stream) {
    MV_CHECK_ARG(x && keys && y, "dropout: NULL pointer");
    MV_CHECK_ARG(B > 0 && per_sample > 0 && per_sample < (1LL << 32) && C > 0 && per_sample % C == 0, "dropout: bad dims B=%d per_sample=%lld C=%d",
                 B, (long long)per_sample, C);
    MV_CHECK_ARG(keep_prob > 0.f && keep_prob <= 1.f, "dropout: keep probability %g outside (0, 1]", keep_prob);
    MV_CHECK_ARG(dtype == MV_BF16 || dtype == MV_F32, "dropout: unknown dtype %d", dtype);
    const long long total = (long long)B * per_sample;
    long long gl = (total + 255) / 256;
    const int grid = (int)(gl > 256 * 16 ? 256 * 16 : gl);
    const bool pair_ok = chw_logical ? (C % 16 == 0) : (per_sample % 16 == 0);
    if (pair_ok && !get_flag("dropout_scalar") && !get_flag("dropout_x8")) {
        const long long pps = per_sample / 16, total_pairs = (long long)B * pps;
        long long gv = (total_pairs + 255) / 256;
        const int gridv = (int)(gv > 256 * 32 ? 256 * 32 : gv);
        set_kernel_name("dropout_threefry_pairs");
        if (dtype == MV_F32)
            hipLaunchKernelGGL(dropout_pair_kernel<float>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                               (const uint32_t*)keys, (float*)y, (long long)per_sample, C, (long long)(per_sample / C), chw_logical,
                               keep_prob, pps, total_pairs);
        else
            hipLaunchKernelGGL(dropout_pair_kernel<bf16_t>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                               (const uint32_t*)keys, (bf16_t*)y, (long long)per_sample, C, (long long)(per_sample / C), chw_logical,
                               keep_prob, pps, total_pairs);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    if (C % 8 == 0 && !get_flag("dropout_scalar")) {
        const long long total8 = total / 8;
        long long gv = (total8 + 255) / 256;
        const int gridv = (int)(gv > 256 * 32 ? 256 * 32 : gv);
        set_kernel_name("dropout_threefry_x8");
        if (dtype == MV_F32)
            hipLaunchKernelGGL(dropout_vec8_kernel<float>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const float*)x,
                               (const uint32_t*)keys, (float*)y, (long long)per_sample, C, (long long)(per_sample / C), chw_logical,
                               keep_prob, total8);
        else
            hipLaunchKernelGGL(dropout_vec8_kernel<bf16_t>, dim3(gridv), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                               (const uint32_t*)keys, (bf16_t*)y, (long long)per_sample, C, (long long)(per_sample / C), chw_logical,
                               keep_prob, total8);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
    set_kernel_name("dropout_threefry");
    if (dtype == MV_F32)
        hipLaunchKernelGGL(dropout_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const uint32_t*)keys,
                           (float*)y, (long long)per_sample, C, (long long)(per_sample / C), chw_logical, keep_prob, total);
    else
        hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const uint32_t*)keys,
                           (bf16_t*)y, (long long)per_sample, C, (long long)(per_sample / C), chw_logical, keep_prob, total);
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

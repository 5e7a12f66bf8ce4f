// Dense-prediction support ops, gfx950: bilinear up-sampling of NHWC feature maps
// (`jax.image.resize(x, shape, method="bilinear")` as used by segmentation/_utils.py:52-58 and deeplabv3.py:66-72) and the
// channel concatenation of the ASPP branches (deeplabv3.py:132-136).  Both are pure data movement: HBM-bound, one pass.
#include "common.h"

namespace mv {

// jax.image.resize, "bilinear" (triangle kernel, half-pixel centres): output pixel o samples the input at
// (o + 0.5) * in / out - 0.5; taps that fall outside the image get weight 0 and the rest is renormalised, which for the 2-tap
// linear kernel is the same as clamping the tap index.  Only out >= in (no antialias window): the callers up-sample.
struct Taps {
    int i0, i1;
    float w1;
};
__device__ __forceinline__ Taps taps_for(int o, int in, int out) {
    const float src = (o + 0.5f) * ((float)in / (float)out) - 0.5f;
    const float f = floorf(src);
    int i0 = (int)f, i1 = i0 + 1;
    const float w1 = src - f;
    i0 = i0 < 0 ? 0 : i0;                      // f = -1 (first half pixel): both taps are pixel 0
    i1 = i1 > in - 1 ? in - 1 : i1;            // last half pixel: both taps are pixel in-1
    return {i0, i1, w1};
}

// one thread per output element; NCHW output: x fastest (coalesced stores, the 2x2 source pixels of neighbouring lanes are the
// same cache lines); NHWC output: channel fastest.
template <typename TI, typename TO, bool NCHW>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const TI* __restrict__ x, TO* __restrict__ y, int B, int h, int w,
                                                              int C, int H, int W) {
    const long long total = (long long)B * C * H * W;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        int b, c, oy, ox;
        long long r = idx;
        if (NCHW) {
            ox = (int)(r % W); r /= W;
            oy = (int)(r % H); r /= H;
            c = (int)(r % C); b = (int)(r / C);
        } else {
            c = (int)(r % C); r /= C;
            ox = (int)(r % W); r /= W;
            oy = (int)(r % H); b = (int)(r / H);
        }
        const Taps ty = taps_for(oy, h, H), tx = taps_for(ox, w, W);
        const TI* base = x + (long long)b * h * w * C + c;
        const float v00 = io<TI>::ld(base + ((long long)ty.i0 * w + tx.i0) * C);
        const float v01 = io<TI>::ld(base + ((long long)ty.i0 * w + tx.i1) * C);
        const float v10 = io<TI>::ld(base + ((long long)ty.i1 * w + tx.i0) * C);
        const float v11 = io<TI>::ld(base + ((long long)ty.i1 * w + tx.i1) * C);
        const float top = v00 + (v01 - v00) * tx.w1, bot = v10 + (v11 - v10) * tx.w1;
        io<TO>::st(y + idx, top + (bot - top) * ty.w1);
    }
}

// NCHW fp32 output, W % 4 == 0: four consecutive x per thread (one 16-byte store; the y taps and the index arithmetic are shared)
template <typename TI>
__global__ __launch_bounds__(256) void resize_bilinear_nchw4_kernel(const TI* __restrict__ x, float* __restrict__ y, int B, int h,
                                                                    int w, int C, int H, int W) {
    const int W4 = W >> 2;
    const long long total = (long long)B * C * H * W4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        long long r = idx;
        const int x4 = (int)(r % W4); r /= W4;
        const int oy = (int)(r % H); r /= H;
        const int c = (int)(r % C), b = (int)(r / C);
        const Taps ty = taps_for(oy, h, H);
        const TI* r0 = x + ((long long)b * h + ty.i0) * w * C + c;
        const TI* r1 = x + ((long long)b * h + ty.i1) * w * C + c;
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Taps tx = taps_for(x4 * 4 + k, w, W);
            const float v00 = io<TI>::ld(r0 + (long long)tx.i0 * C), v01 = io<TI>::ld(r0 + (long long)tx.i1 * C);
            const float v10 = io<TI>::ld(r1 + (long long)tx.i0 * C), v11 = io<TI>::ld(r1 + (long long)tx.i1 * C);
            const float top = v00 + (v01 - v00) * tx.w1, bot = v10 + (v11 - v10) * tx.w1;
            o[k] = top + (bot - top) * ty.w1;
        }
        *(float4*)(y + idx * 4) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// rows x row_bytes strided copy in 16-byte (or, for odd sizes, 2-byte) pieces
template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const char* __restrict__ src, char* __restrict__ dst, long long rows,
                                                        int pieces, long long src_pitch, long long dst_pitch) {
    const long long total = rows * pieces;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long r = idx / pieces;
        const int p = (int)(idx - r * pieces);
        *(T*)(dst + r * dst_pitch + (long long)p * sizeof(T)) = *(const T*)(src + r * src_pitch + (long long)p * sizeof(T));
    }
}

static int grid_of(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g > 256 * 16 ? 256 * 16 : (g < 1 ? 1 : g));
}

}  // namespace mv

using namespace mv;

extern "C" {

int mv_resize_bilinear_nhwc_fwd(const void* x, void* y, int N, int h, int w, int C, int H, int W, int in_dtype, int out_dtype,
                                int out_nchw, mv_stream_t stream) {
    MV_CHECK_ARG(x && y && N > 0 && h > 0 && w > 0 && C > 0 && H > 0 && W > 0, "resize_bilinear: bad args");
    if (H < h || W < w) {
        set_error("resize_bilinear: down-sampling (%dx%d -> %dx%d) needs the antialiasing window of jax.image.resize; "
                  "only up-sampling is on the path (segmentation/_utils.py:52-58)", h, w, H, W);
        return MV_E_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)N * C * H * W;
    set_kernel_name(out_nchw ? "resize_bilinear_nhwc_to_nchw" : "resize_bilinear_nhwc");
    if (out_nchw && out_dtype == MV_F32 && W % 4 == 0 && ((uintptr_t)y % 16 == 0)) {
        if (in_dtype == MV_BF16)
            hipLaunchKernelGGL((resize_bilinear_nchw4_kernel<bf16_t>), dim3(grid_of(total / 4)), dim3(256), 0, st, (const bf16_t*)x,
                               (float*)y, N, h, w, C, H, W);
        else
            hipLaunchKernelGGL((resize_bilinear_nchw4_kernel<float>), dim3(grid_of(total / 4)), dim3(256), 0, st, (const float*)x,
                               (float*)y, N, h, w, C, H, W);
        MV_LAUNCH_CHECK();
        return MV_OK;
    }
#define GO(TI, TO)                                                                                                        \
    do {                                                                                                                  \
        if (out_nchw)                                                                                                     \
            hipLaunchKernelGGL((resize_bilinear_kernel<TI, TO, true>), dim3(grid_of(total)), dim3(256), 0, st, (const TI*)x, \
                               (TO*)y, N, h, w, C, H, W);                                                                 \
        else                                                                                                              \
            hipLaunchKernelGGL((resize_bilinear_kernel<TI, TO, false>), dim3(grid_of(total)), dim3(256), 0, st, (const TI*)x, \
                               (TO*)y, N, h, w, C, H, W);                                                                 \
    } while (0)
    if (in_dtype == MV_BF16 && out_dtype == MV_F32) GO(bf16_t, float);
    else if (in_dtype == MV_BF16) GO(bf16_t, bf16_t);
    else if (out_dtype == MV_F32) GO(float, float);
    else GO(float, bf16_t);
#undef GO
    MV_LAUNCH_CHECK();
    return MV_OK;
}

int mv_copy_rows(const void* src, void* dst, int64_t rows, int64_t row_bytes, int64_t src_pitch, int64_t dst_pitch,
                 mv_stream_t stream) {
    MV_CHECK_ARG(src && dst && rows > 0 && row_bytes > 0 && src_pitch >= row_bytes && dst_pitch >= row_bytes, "copy_rows: bad args");
    MV_CHECK_ARG(row_bytes % 2 == 0 && src_pitch % 2 == 0 && dst_pitch % 2 == 0, "copy_rows: sizes must be multiples of 2 bytes");
    hipStream_t st = (hipStream_t)stream;
    set_kernel_name("copy_rows");
    const bool v16 = row_bytes % 16 == 0 && src_pitch % 16 == 0 && dst_pitch % 16 == 0 && ((uintptr_t)src % 16 == 0) &&
                     ((uintptr_t)dst % 16 == 0);
    if (v16) {
        const int pieces = (int)(row_bytes / 16);
        hipLaunchKernelGGL((copy_rows_kernel<uint4>), dim3(grid_of(rows * pieces)), dim3(256), 0, st, (const char*)src, (char*)dst,
                           (long long)rows, pieces, (long long)src_pitch, (long long)dst_pitch);
    } else {
        const int pieces = (int)(row_bytes / 2);
        hipLaunchKernelGGL((copy_rows_kernel<uint16_t>), dim3(grid_of(rows * pieces)), dim3(256), 0, st, (const char*)src,
                           (char*)dst, (long long)rows, pieces, (long long)src_pitch, (long long)dst_pitch);
    }
    MV_LAUNCH_CHECK();
    return MV_OK;
}

}  // extern "C"

// Shared device/host helpers for the gfx950 kernels.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/eqxvision_amd.h"

namespace mv {

typedef uint16_t bf16_t;  // raw bf16 bits

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even fp32 -> bf16: plain casts, which hipcc lowers to the gfx950 hardware converter
// (v_cvt_pk_bf16_f32, two values per instruction) instead of ~6 VALU ops of bit arithmetic per value
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    bf16x2_t v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct io;
template <> struct io<float> {
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct io<bf16_t> {
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

__device__ __forceinline__ float gelu_tanh_f(float x) {
    // jax.nn.gelu(approximate=True): 0.5x(1+tanh(u)), u = sqrt(2/pi)(x+0.044715x^3)
    // 0.5(1+tanh(u)) = 1/(1+exp(-2u)), exp(-2u) = exp2(x*(k0 + k1*x^2)): one v_exp_f32 + one v_rcp_f32 (hardware
    // approximations, ~1 ulp) and four plain VALU ops; no IEEE division, no clamp (exp2 -> +inf gives rcp -> 0 -> -0, the
    // limit for very negative x; exp2 -> 0 gives x).  The epilogue of the 768->3072 fc1 GEMM spent ~28% of the layer in the
    // first version of this function, and the fused Swin MLP kernel (ln_mlp.hip) is bound by it.
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2*sqrt(2/pi)*log2(e)
    const float k1 = k0 * 0.044715f;
    const float t = x * fmaf(k1, x * x, k0);
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

// Two activations at once on the packed-fp32 VALU ops (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 take two values per
// lane per issue slot); the two transcendentals per value stay scalar.  Returns the bf16 pair.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t gelu_tanh_pack2(float a, float b) {
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float k1 = k0 * 0.044715f;
    f32x2_t x = {a, b};
    const f32x2_t kk0 = {k0, k0}, kk1 = {k1, k1}, one = {1.0f, 1.0f};
    f32x2_t t = x * __builtin_elementwise_fma(kk1, x * x, kk0);
    f32x2_t e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    e = e + one;
    f32x2_t r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    x = x * r;
    return pack_bf2(x[0], x[1]);
}

template <int ACT> __device__ __forceinline__ float apply_act(float v) {
    if (ACT == MV_ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == MV_ACT_GELU_TANH) return gelu_tanh_f(v);
    return v;
}
__device__ __forceinline__ float apply_act_rt(float v, int act) {
    if (act == MV_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MV_ACT_GELU_TANH) return gelu_tanh_f(v);
    if (act == MV_ACT_HARD_SWISH) return v * fminf(fmaxf(v + 3.0f, 0.f), 6.0f) * (1.0f / 6.0f);       // jax.nn.hard_swish
    if (act == MV_ACT_HARD_SIGMOID) return fminf(fmaxf(v + 3.0f, 0.f), 6.0f) * (1.0f / 6.0f);         // jax.nn.hard_sigmoid
    if (act == MV_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
    if (act == MV_ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
    return v;
}

// ---- host side ------------------------------------------------------------------------
void set_error(const char* fmt, ...);
void set_kernel_name(const char* name);
void append_kernel_name(const char* suffix);
int get_flag(const char* name);
// the scratch the host handed over with mv_set_scratch for the next launch on `stream`: returned (and forgotten) if it holds
// at least `need` bytes, else nullptr
constexpr size_t SCRATCH_SYNC_BYTES = 4096;      // head of every scratch offer: split-K arrival words, zero between launches
void* take_scratch(hipStream_t stream, size_t need);
void* peek_scratch(hipStream_t stream, size_t* bytes);
size_t splitk_scratch_bytes(long long M, long long N, long long kred);       // igemm8.hip
const void* zero_page(hipStream_t stream);  // >= 4 KiB of device zeros for the current device

inline size_t dsize(int dt) { return dt == MV_BF16 ? 2 : 4; }

#define MV_CHECK_ARG(cond, ...)                      \
    do {                                             \
        if (!(cond)) {                               \
            mv::set_error(__VA_ARGS__);              \
            return MV_E_INVALID;                     \
        }                                            \
    } while (0)

// the matrix-core epilogues implement none / relu / gelu only
#define MV_CHECK_FUSED_ACT(act, who) \
    MV_CHECK_ARG((act) >= MV_ACT_NONE && (act) <= MV_ACT_GELU_TANH, who ": activation %d is not fused into this entry (use mv_eltwise_fwd)", (act))

#define MV_LAUNCH_CHECK()                                                            \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            mv::set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return (int)e__;                                                         \
        }                                                                            \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE property of a kernel: a launch site keeps one of these (static) and asks
// before every launch; the attribute is set the first time the site launches on a device (and again if it needs more LDS than any
// earlier launch asked for on that device).  Lock-free: two threads racing on the first launch both set the same value.
struct LdsAttrSite {
    std::atomic<int> bytes[16];                 // per device ordinal (mod 16): the largest size already granted
    LdsAttrSite() { for (auto& b : bytes) b.store(0, std::memory_order_relaxed); }
    hipError_t ensure(const void* kern, size_t need) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::atomic<int>& b = bytes[dev & 15];
        if (b.load(std::memory_order_acquire) >= (int)need) return hipSuccess;
        e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
        if (e == hipSuccess) b.store((int)need, std::memory_order_release);
        return e;
    }
};

#define MV_HIP(call)                                                                 \
    do {                                                                             \
        hipError_t e__ = (call);                                                     \
        if (e__ != hipSuccess) {                                                     \
            (void)hipGetLastError(); /* clear the sticky error so later launches are not blamed */ \
            mv::set_error("%s:%d %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
            return (int)e__;                                                         \
        }                                                                            \
    } while (0)

// dispatch helpers implemented in the kernel translation units
int igemm_supported(int C, int K, int R, int S, int groups, int in_dtype, int out_dtype);
int igemm_launch(const void* x, const void* w, const float* scale, const float* shift, const void* residual,
                 void* y, int N, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw,
                 int dh, int dw, int act, int in_dtype, int out_dtype, hipStream_t stream);
int igemm_grouped64_supported(int C, int K, int R, int S, int groups, int in_dtype, int out_dtype);
int igemm_grouped64_window(int C, int groups);
int igemm_grouped64_launch(const void* x, const void* w64, const float* scale, const float* shift, const void* residual, void* y,
                           int N, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw, int groups,
                           int act, int out_dtype, hipStream_t stream);
int igemm_oddc_supported(int C, int K, int groups, int in_dtype, int out_dtype);
int igemm_oddc_launch(const void* x, const void* w, const float* scale, const float* shift, const void* residual, void* y, int N,
                      int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw, int act,
                      int out_dtype, hipStream_t stream);
int dwconv_supported(int C, int K, int groups, int R, int S, int in_dtype, int out_dtype);
int dwconv_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int H, int W, int C, int R,
                  int S, int sh, int sw, int ph, int pw, int dh, int dw, int act, hipStream_t stream);
int stream1x1_supported(int C, int K, int in_dtype, int out_dtype, long long M);
int stream1x1_launch(const void* x, const void* w, const float* scale, const float* shift, const void* residual,
                     void* y, long long M, int C, int K, int act, int out_dtype, hipStream_t stream);
int ln_mlp_supported(long long M, int C, int hidden, int x_dtype);
int ln_mlp_launch(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, long long M,
                  float eps, int x_dtype, hipStream_t st);
int stream1x1_ln_supported(long long M, int C, int K, int x_dtype, int out_dtype);
int stream1x1_ln_launch(const void* x, const void* w, const float* shift, void* y, long long M, int C, int K, float eps, int act,
                        int x_dtype, hipStream_t st);
int conv3x3c64_supported(int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw, int in_dtype,
                         int out_dtype, const void* residual, long long M);
int igemm2_wanted(long long M, int C, int K, int R, int S);
int igemm2_launch(const void* x, const void* w, const float* scale, const float* shift, const void* residual,
                  void* y, int N, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw,
                  int dh, int dw, int act, int out_dtype, int m_end, int tok, hipStream_t stream);
int igemm2_tile_shape(long long M, int K, int* bm, int* bn);
int igemm2_dual_supported(long long M, int C1, int C2, int K, int dtype);
int igemm2_dual_launch(const void* x, const void* x2, const void* w, const float* scale, const float* shift, void* y, int N,
                       int Ho, int Wo, int C1, int H2, int W2, int C2, int s2, int K, int act, hipStream_t st);
void igemm2_force_tile(int tile);
int tile_override(const char* kind, long long M, int a, int b, int R, int S, int sh);
int chain1x1_supported(long long M, int C, int K, int N2, int dtype);
int chain1x1_dual_supported(long long M, int C1, int C2, int K, int N2, int dtype);
int chain1x1_dual_launch(const void* x, const void* x2, const void* wcat, const float* scale3, const float* shift3, void* y,
                         const void* w1, const float* scale1, const float* shift1, void* t1, long long M, hipStream_t st);
int chain1x1_launch(const void* x, const void* w3, const float* scale3, const float* shift3, const void* residual, void* y,
                    const void* w1, const float* scale1, const float* shift1, void* t1, long long M, int N2, hipStream_t st);
int chain1x1_sub_supported(long long N, int H, int W, int C, int K, int N2, int dtype);
int chain1x1_sub_launch(const void* x, const void* w3, const float* scale3, const float* shift3, const void* residual, void* y_sub,
                        const void* w1, const float* scale1, const float* shift1, void* t1, int N, int H, int W, hipStream_t st);
int chain_rc_supported(long long M, int C, int K, int N2, int dtype);
int chain_rc_launch(const void* t2, const void* t2_prev, const void* x0, const void* wfrag, const void* shifts, void* y, void* t1,
                    long long M, hipStream_t st);
int chain_rc0_launch(const void* t2, const void* x0, const void* wfrag, const void* shifts, void* t1, long long M, hipStream_t st);
int chain_res_supported(long long N, int H, int W, int C, int K, int N2, int sub, int dtype);
int chain_res_launch(const void* t2, const void* residual, const void* wfrag, const void* shifts, void* y, void* t1, int N, int H, int W,
                     int sub, hipStream_t st);
int chain_stream_supported(long long M, int C, int K, int N2, int dtype);
int chain_stream_launch(const void* x, const void* w3, const float* scale3, const float* shift3, const void* residual, void* y,
                        const void* w1, const float* scale1, const float* shift1, void* t1, long long M, hipStream_t st);
int igemm8_supported(long long M, int C, int K, int R, int S, long long x_bytes, long long w_bytes);
int device_status(int clear, unsigned* out);              // igemm8.hip: the status word of mv_device_status
int igemm8_wanted(long long M, int C, int K, int R, int S);
int igemm8_launch(const void* x, const void* w, const float* scale, const float* shift, const void* residual, void* y,
                  int N, int H, int W, int C, int K, int R, int S, int sh, int sw, int ph, int pw, int dh, int dw,
                  int act, int out_dtype, int tok, int tile, hipStream_t st);   // tile: 1 = 256x256, 2 = 128x256, 3 = 256x128
bool igemm8_ln_supported(long long M, int N, int K);
int igemm8_lnout_launch(const void* x, const void* w, const float* shift, const void* res, const void* res_lo, void* y, void* y_lo,
                        float* stats, long long M, int N, int K, hipStream_t st);
int igemm8_lnin_launch(const void* x, const float* stats, const void* w, const float* colsum, const float* shift, void* y, long long M,
                       int N, int K, float eps, int act, int tok, hipStream_t st);
int igemm8_dual_launch(const void* x, const void* x2, const void* w, const float* scale, const float* shift,
                       const void* residual, void* y, int N, int Ho, int Wo, int C1, int H2, int W2, int C2, int s2, int K,
                       int act, int out_dtype, int tile, hipStream_t st);
int stem_supported(int C, int K, int R, int S, int x_dtype, int out_dtype);
int stem_f32out_supported(int C, int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int x_dtype);
int stem_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int C,
                int H, int W, int K, int R, int S, int sh, int sw, int ph, int pw, int act, int x_dtype,
                int out_dtype, int tok_stride, int tok_offset, const float* pos, hipStream_t stream,
                const void* w_lo = nullptr);   // w_lo: low halves of split-precision weights (generic kernel only)
int skinny_f32_supported(long long M, int C, int K, int in_dtype, int out_dtype, const void* residual);
int skinny_f32_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, long long M, int C, int K,
                      int act, hipStream_t stream);
int swin_mfma_supported(int C, int heads, int ws_h, int ws_w, int dtype);
int swin_mfma_launch(const void* qkv, const float* bias, void* out, int B, int Hf, int Wf, int C, int heads, int ws_h,
                     int ws_w, int shift_h, int shift_w, const uint32_t* drop_keys, float keep, hipStream_t st);
int conv3x3c64_v2_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int H,
                         int W, int act, hipStream_t stream);
int skinny_supported(long long M, int C, int K, int in_dtype, const void* residual);
int skinny_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, long long M, int C, int K,
                  int act, int out_dtype, hipStream_t stream);
int stem_pool_supported(int C, int K, int R, int S, int sh, int sw, int ph, int pw, int pk, int ps, int pp, int act,
                        int x_dtype, int out_dtype, long long in_elems);
int stem_pool_launch(const void* x, const void* w, const float* scale, const float* shift, void* y, int N, int H, int W, int R,
                     int x_dtype, hipStream_t stream);
int mha_mfma_supported(int N, int dh, int dtype);
int mha_mfma_launch(const void* qkv, int head_major, void* out, float* probs, int B, int N, int H, int dh, float scale,
                    const uint32_t* drop_keys, float keep, hipStream_t st);

}  // namespace mv

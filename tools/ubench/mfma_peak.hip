// MFMA ceiling microbenchmark: register-resident v_mfma_f32_32x32x16_bf16 loops, no memory traffic.
// usage: ./mfma_peak   (prints TFLOP/s for 1, 2, 4 waves per SIMD, zero vs random operands)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(1024) void mfma_loop(const uint4* in, float* out, int iters) {
    uint4 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][7];
    if (s == 12345.678f) out[0] = s;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(1024) void mfma_loop16(const uint4* in, float* out, int iters) {
    uint4 a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    uint4* d; float* o;
    hipMalloc(&d, 128 * 16); hipMalloc(&o, 4);
    for (int rnd = 0; rnd < 2; ++rnd) {
        unsigned short h[128 * 8];
        for (int i = 0; i < 128 * 8; ++i) { float f = rnd ? ((rand() % 2001) - 1000) / 1000.0f : 0.f; unsigned u; memcpy(&u, &f, 4); h[i] = u >> 16; }
        hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        for (int wpb : {256, 512, 1024}) {      // threads per block: 1, 2, 4 waves per SIMD with 1 block per CU
            const int iters = 4000, blocks = 256;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(wpb), 0, 0, d, o, iters);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(wpb), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            double flop = (double)blocks * (wpb / 64) * iters * 4 * 32768.0;
            printf("32x32x16 %s operands, %d waves/SIMD: %.1f TFLOP/s (%.3f ms)\n", rnd ? "random" : "zero  ", wpb / 256, flop / ms / 1e9, ms);
            hipLaunchKernelGGL(mfma_loop16<8>, dim3(blocks), dim3(wpb), 0, 0, d, o, iters);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(mfma_loop16<8>, dim3(blocks), dim3(wpb), 0, 0, d, o, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            flop = (double)blocks * (wpb / 64) * iters * 8 * 16384.0;
            printf("16x16x32 %s operands, %d waves/SIMD: %.1f TFLOP/s (%.3f ms)\n", rnd ? "random" : "zero  ", wpb / 256, flop / ms / 1e9, ms);
        }
    }
    return 0;
}

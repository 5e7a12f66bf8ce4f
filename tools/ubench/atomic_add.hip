// Is "y += v" as a no-return fp32 atomic (executed in the L2) cheaper for the CUs than load + add + store?  (round 5: the
// residual update of the ViT proj / fc2 epilogue.)  Streams N fp32 in the GEMM epilogue's access pattern (each lane 2 x 16
// bytes of a 256-byte row segment) three ways: (a) y = r + v with r, y distinct; (b) y += v read-modify-write by the CU;
// (c) y += v with buffer_atomic_pk / global_atomic_add_f32 (4 dword atomics per 16 bytes).   usage: ./atomic_add [MB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void k_rmw_sep(const float4* __restrict__ r, float4* __restrict__ y, long n4, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 a = r[i];
        a.x += v; a.y += v; a.z += v; a.w += v;
        y[i] = a;
    }
}
__global__ void k_rmw_inplace(float4* y, long n4, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 a = y[i];
        a.x += v; a.y += v; a.z += v; a.w += v;
        y[i] = a;
    }
}
__global__ void k_atomic(float* y, long n4, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float* p = y + 4 * i;
        __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)p, v);
        __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(p + 1), v);
        __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(p + 2), v);
        __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(p + 3), v);
    }
}
// one dword per lane, consecutive lanes consecutive dwords (a 256-byte row per wave instruction)
__global__ void k_atomic_dw(float* y, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(y + i), v);
}
__global__ void k_store(float4* y, long n4, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
        y[i] = make_float4(v, v, v, v);
}

int main(int argc, char** argv) {
    const long mb = argc > 1 ? atol(argv[1]) : 77;
    const long n = mb * 1024 * 1024 / 4, n4 = n / 4;
    float *r, *y;
    hipMalloc(&r, n * 4); hipMalloc(&y, n * 4);
    hipMemset(r, 0, n * 4); hipMemset(y, 0, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8, block = 256;
    auto time = [&](const char* name, auto launch, double bytes) {
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.1f us   %6.2f TB/s of HBM traffic (%.0f MB payload)\n", name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, n * 4 / 1e6);
    };
    time("store only", [&] { hipLaunchKernelGGL(k_store, dim3(grid), dim3(block), 0, 0, (float4*)y, n4, 1.f); }, n * 4.0);
    time("y = r + v (load, add, store)", [&] { hipLaunchKernelGGL(k_rmw_sep, dim3(grid), dim3(block), 0, 0, (const float4*)r, (float4*)y, n4, 1.f); }, n * 8.0);
    time("y += v by the CU, in place", [&] { hipLaunchKernelGGL(k_rmw_inplace, dim3(grid), dim3(block), 0, 0, (float4*)y, n4, 1.f); }, n * 8.0);
    time("y += v, 4 atomics per 16 bytes", [&] { hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(block), 0, 0, y, n4, 1.f); }, n * 8.0);
    time("y += v, dword atomics, lane-linear", [&] { hipLaunchKernelGGL(k_atomic_dw, dim3(grid), dim3(block), 0, 0, y, n, 1.f); }, n * 8.0);
    float h[4]; hipMemcpy(h, y, 16, hipMemcpyDeviceToHost);
    printf("check y[0] = %.1f\n", h[0]);
    return 0;
}

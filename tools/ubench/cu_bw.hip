// How much HBM bandwidth can a SUBSET of the CUs pull?  (round 5: could the HBM-bound kernels of a forward run on part of the chip
// while matrix kernels use the rest?)  A read-only / copy stream over 1 GB with `grid` workgroups of 1024 threads (one per CU while
// grid <= 256), 8 x 16-byte loads in flight per lane.   usage: ./cu_bw
#include <hip/hip_runtime.h>
#include <stdio.h>

template <bool COPY>
__global__ __launch_bounds__(1024) void stream(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16, unsigned* sink) {
    const long stride = (long)gridDim.x * 1024;
    long i = (long)blockIdx.x * 1024 + threadIdx.x;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (; i + 7 * stride < n16; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (COPY) dst[i + u * stride] = v[u];
            else { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    }
    if (!COPY && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

int main() {
    const long bytes = 1L << 30, n16 = bytes / 16;
    uint4 *a, *b; unsigned* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int copy = 0; copy < 2; ++copy)
        for (int grid : {16, 32, 64, 96, 128, 192, 256, 512}) {
            for (int it = 0; it < 2; ++it) {
                hipEventRecord(e0);
                if (copy) hipLaunchKernelGGL(stream<true>, dim3(grid), dim3(1024), 0, 0, a, b, n16, sink);
                else hipLaunchKernelGGL(stream<false>, dim3(grid), dim3(1024), 0, 0, a, b, n16, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (it == 1) printf("%s grid %3d: %7.1f us  %5.2f TB/s  (%5.1f GB/s per workgroup)\n", copy ? "copy" : "read", grid, ms * 1e3,
                                    (copy ? 2.0 : 1.0) * bytes / (ms * 1e-3) / 1e12, (copy ? 2.0 : 1.0) * bytes / (ms * 1e-3) / 1e9 / grid);
            }
        }
    return 0;
}

// HBM read efficiency for G-byte granules at a fixed row stride (the access pattern of per-head K/V staging
// from a [B, N, 3, H, dh] qkv tensor).  hipcc --offload-arch=gfx950 -O3 stride_read.hip -o stride_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

template <int UNROLL>
__global__ __launch_bounds__(256) void rd(const uint4* __restrict__ x, uint4* __restrict__ sink, long long rows,
                                          int gran16, int stride16, int ngran) {
    // work item = (granule column g, row r): lanes cover gran16 16-byte chunks of a granule, then next rows
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long nthreads = (long long)gridDim.x * blockDim.x;
    const long long items = rows * gran16;            // per granule column
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int g = 0; g < ngran; ++g) {
        for (long long i = tid; i < items; i += nthreads * UNROLL) {
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                long long j = i + u * nthreads;
                j = j < items ? j : items - 1;
                const long long r = j / gran16, c = j - r * gran16;
                v[u] = x[r * stride16 + (long long)g * gran16 + c];
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int stride = 4608;                       // bytes per row (3 * 768 bf16)
    const long long rows = 256LL * 197;
    size_t bytes = (size_t)rows * stride;
    uint4 *x, *sink;
    hipMalloc(&x, bytes); hipMalloc(&sink, 64);
    hipMemset(x, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grans[] = {128, 256, 512, 1536, 4608};
    for (int gi = 0; gi < 5; ++gi) {
        const int G = grans[gi];
        const int ngran = stride / G;
        for (int blocks : {1024, 4096}) {
            float best = 1e9f;
            for (int it = 0; it < 5; ++it) {
                hipEventRecord(e0);
                // each launch reads granule columns [0, ngran): the whole tensor once, column by column
                hipLaunchKernelGGL(rd<4>, dim3(blocks), dim3(256), 0, 0, x, sink, rows, G / 16, stride / 16, ngran);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("granule %5d B  blocks %5d: %8.1f us  %7.1f GB/s\n", G, blocks, best * 1e3, bytes / best / 1e6);
        }
    }
    return 0;
}

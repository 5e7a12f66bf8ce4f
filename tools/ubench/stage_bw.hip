// L2 -> LDS staging throughput per CU on gfx950: LDS-DMA (global_load_lds_dwordx4) vs register staging
// (global_load_dwordx4 + ds_write_b128), for the row shapes the implicit-GEMM kernels stage.
//   hipcc --offload-arch=gfx950 -O3 stage_bw.hip -o stage_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// MODE 0: LDS-DMA; MODE 1: registers + ds_write.  ROWB = bytes per source row (64 / 128 / 1024 = contiguous).
template <int MODE, int ROWB, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void stage(const char* __restrict__ src, long long pitch, long long win, int iters, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int LPR = ROWB / 16;                 // lanes per row
    constexpr int RPI = 64 / LPR;                  // rows per instruction
    // each wave owns 4 pieces per iteration, ring of 4 stages x WAVES x 4 KB
    // WIN bytes of source are cycled through: 16 KB = L1 hits, 2 MB = L2 hits (L1 misses), 48 MB = MALL / HBM
    const long long lane_off = (long long)(lane / LPR) * pitch + (lane % LPR) * 16;
    const long long nslots = win / 16384;
    char* lbase = smem + wave * 4096;
    uint4 r[4];
    for (int it = 0; it < iters; ++it) {
        const char* g = src + (((long long)blockIdx.x * 7 + it) % nslots) * 16384 + lane_off;
        char* l = lbase + (it & 3) * (WAVES * 4096);
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16(g + (long long)q * RPI * pitch, l + q * 1024);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            if (it > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int q = 0; q < 4; ++q) *(uint4*)(l + q * 1024 + lane * 16) = r[q];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint4* gp = (const uint4*)(g + (long long)q * RPI * pitch);
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[q]) : "v"(gp) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (((int*)smem)[tid] == 0x12345678) sink[0] = 1;
}

template <int MODE, int ROWB, int WAVES>
static void run(const char* name, const char* src, int* sink, long long win) {
    const int iters = 4096;
    const size_t smem = (size_t)4 * WAVES * 4096;
    auto k = stage<MODE, ROWB, WAVES>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), smem, 0, src, (long long)(ROWB == 1024 ? 1024 : 1536), win, iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes = 256.0 * WAVES * 4096.0 * iters;
    printf("%-44s %8.1f us  %7.2f TB/s  %6.1f GB/s per CU\n", name, best * 1e3, bytes / best / 1e9, bytes / best / 1e6 / 256);
}

int main() {
    char* src; int* sink;
    hipMalloc(&src, 64 << 20); hipMemset(src, 1, 64 << 20); hipMalloc(&sink, 64);
    const long long wins[3] = {16384, 2 << 20, 48 << 20};
    const char* wn[3] = {"L1-resident", "L2-resident", "48 MB window"};
    for (int w = 0; w < 3; ++w) {
        printf("-- source window: %s\n", wn[w]);
        run<0, 64, 8>("LDS-DMA   64-B rows   8 waves", src, sink, wins[w]);
        run<0, 128, 8>("LDS-DMA  128-B rows   8 waves", src, sink, wins[w]);
        run<0, 1024, 8>("LDS-DMA  contiguous   8 waves", src, sink, wins[w]);
        run<0, 128, 4>("LDS-DMA  128-B rows   4 waves", src, sink, wins[w]);
        run<1, 64, 8>("regs+ds_write  64-B rows   8 waves", src, sink, wins[w]);
        run<1, 128, 8>("regs+ds_write 128-B rows   8 waves", src, sink, wins[w]);
        run<1, 1024, 8>("regs+ds_write contiguous   8 waves", src, sink, wins[w]);
    }
    return 0;
}

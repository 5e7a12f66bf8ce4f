// L2-RESIDENT stream rate per CU versus bytes in flight (round 6, review item 3).
//
// Three no-gos of round 5 (bneck_tail for ResNet layer 4, a deeper weight ring for Swin stage 2, the 64-row ViT MLP) priced a
// workgroup's WEIGHT stream at the 57 GB/s that tools/ubench/cu_bw.hip measured for an HBM stream with 128 KB in flight.  A weight
// stream is not an HBM stream: every workgroup re-reads the SAME 1-8 MB, so after the first pass the bytes come out of the XCD's
// 4 MB L2 (or the 256 MB Infinity Cache).  This probe measures that: every workgroup (8 waves = 512 lanes, the shape of the
// forward's big kernels; one workgroup per CU while grid <= 256) streams the same `buf_mb` buffer `passes` times
//   * through REGISTERS : D x 16-byte global loads in flight per lane  (D * 8 KB in flight per workgroup)
//   * through LDS-DMA   : D x `buffer_load_dwordx4 ... lds` in flight per lane, counted `s_waitcnt vmcnt` (a ring of D x 8 KB in LDS)
// and prints GB/s per workgroup and TB/s in aggregate.   usage: ./l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(512) void reg_stream(const uint4* __restrict__ src, long n16, int passes, unsigned* sink) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    const long chunk = 512L * D;                                  // uint4s per workgroup step
    const long start = ((long)blockIdx.x * 7919 * chunk) % n16;   // workgroups start at different places: no lock-step hot line
    for (int p = 0; p < passes; ++p) {
        long base = start;
        for (long done = 0; done + chunk <= n16; done += chunk) {
            uint4 v[D];
#pragma unroll
            for (int u = 0; u < D; ++u) v[u] = src[base + u * 512 + threadIdx.x];
#pragma unroll
            for (int u = 0; u < D; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
            base += chunk;
            if (base + chunk > n16) base = 0;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

// D-deep software pipeline: D loads issued, then per step wait for the oldest and issue one more (the ring depth stays D)
template <int D>
__global__ __launch_bounds__(512) void reg_ring(const uint4* __restrict__ src, long n16, int passes, unsigned* sink) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    const long steps = n16 / 512;
    const long start = ((long)blockIdx.x * 7919) % steps;
    uint4 v[D];
    for (int p = 0; p < passes; ++p) {
        long s = start;
#pragma unroll
        for (int u = 0; u < D; ++u) { v[u] = src[s * 512 + threadIdx.x]; s = s + 1 == steps ? 0 : s + 1; }
        for (long done = D; done + D <= steps; done += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w;
                v[u] = src[s * 512 + threadIdx.x];
                s = s + 1 == steps ? 0 : s + 1;
            }
        }
#pragma unroll
        for (int u = 0; u < D; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}

__device__ __forceinline__ u32x4 make_rsrc(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r[2] = 0x80000000u;
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void dma16(unsigned ldsw, unsigned voff, const u32x4& rsrc, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(ldsw), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS-DMA ring of D slots x 8 KB (512 lanes x 16 bytes); nothing reads the LDS (the probe measures the load path, not the consumer)
template <int D>
__global__ __launch_bounds__(512) void dma_ring(const uint4* __restrict__ src, long n16, int passes, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const u32x4 rs = make_rsrc(src);
    const unsigned voff = (unsigned)(wave * 64 + lane) * 16u;
    const long steps = n16 / 512;                                  // 8 KB steps in the buffer (< 2^31 bytes)
    const long start = ((long)blockIdx.x * 7919) % steps;
    for (int p = 0; p < passes; ++p) {
        long s = start;
#pragma unroll
        for (int u = 0; u < D; ++u) {
            dma16(__builtin_amdgcn_readfirstlane(lds0 + u * 8192 + wave * 1024), voff, rs, (unsigned)(s * 8192));
            s = s + 1 == steps ? 0 : s + 1;
        }
        for (long done = D; done + D <= steps; done += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                wait_vm<D - 1>();                                   // slot u's previous load has landed
                dma16(__builtin_amdgcn_readfirstlane(lds0 + u * 8192 + wave * 1024), voff, rs, (unsigned)(s * 8192));
                s = s + 1 == steps ? 0 : s + 1;
            }
        }
        wait_vm<0>();
    }
    if (smem[threadIdx.x] == 77 && passes < 0) *sink = 1;
}

template <typename K>
static void run(const char* how, K kern, int depth, int lds, int grid, const uint4* buf, long bytes, int passes, unsigned* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, buf, bytes / 16, passes, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    const double moved = (double)bytes * passes * grid;
    printf("%-9s %3d KB in flight  grid %3d  buffer %2ld MB: %8.1f us  %6.1f GB/s per workgroup  %6.2f TB/s aggregate\n", how, depth * 8,
           grid, bytes >> 20, best * 1e3, moved / (best * 1e-3) / 1e9 / grid, moved / (best * 1e-3) / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

#define REG(Dv)  run("registers", reg_ring<Dv>, Dv, 0, grid, buf, bytes, passes, sink)
#define DMA(Dv)                                                                                                            \
    do {                                                                                                                   \
        hipFuncSetAttribute((const void*)dma_ring<Dv>, hipFuncAttributeMaxDynamicSharedMemorySize, Dv * 8192);             \
        run("lds-dma", dma_ring<Dv>, Dv, Dv * 8192, grid, buf, bytes, passes, sink);                                        \
    } while (0)

int main() {
    uint4* buf; unsigned* sink;
    hipMalloc(&buf, 64L << 20); hipMalloc(&sink, 4);
    hipMemset(buf, 1, 64L << 20);
    for (long mb : {1L, 2L, 4L, 8L, 32L}) {
        const long bytes = mb << 20;
        const int passes = (int)((256L << 20) / bytes / 4);          // 64 MB per workgroup
        for (int grid : {64, 128, 256}) {
            REG(2); REG(4); REG(8); REG(16); REG(24);
            DMA(2); DMA(4); DMA(8); DMA(16);
            if (grid == 256) { DMA(12); }
            printf("\n");
        }
    }
    // the straight batch form (D loads, then D uses) at the depth the forward's register rings use, for comparison with the ring
    {
        const long bytes = 2L << 20; const int passes = 32, grid = 256;
        run("reg-batch", reg_stream<8>, 8, 0, grid, buf, bytes, passes, sink);
        run("reg-batch", reg_stream<16>, 16, 0, grid, buf, bytes, passes, sink);
    }
    return 0;
}

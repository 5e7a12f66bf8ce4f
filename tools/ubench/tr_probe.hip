// Empirical semantics of ds_read_b64_tr_b16 on gfx950: LDS holds ushort[i] = i; every lane supplies a byte address,
// the 4 returned 16-bit elements per lane are dumped.   hipcc --offload-arch=gfx950 -O3 tr_probe.hip -o tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = l * 8;                                   // consecutive 8-byte chunks
    else if (mode == 1) addr = (l & 15) * 2 + (l >> 4) * 128;      // the guide's formula as a per-lane address
    else if (mode == 2) addr = (l & 15) * 64 + (l >> 4) * 8;       // rows of 64 B (32 elements), 16 rows per group
    else addr = (l & 3) * 8 + ((l >> 2) & 3) * 64 + (l >> 4) * 256; // 4x4 blocks: 4 lanes along a row, 4 rows of 64 B
    addr += (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned short*)lds;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (element indices = byte address / 2)\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}

// Shared by the deep-pipelined implicit-GEMM kernels (igemm2.hip, igemm8.hip): launch parameters, inline-asm
// LDS / waitcnt primitives (invisible to hipcc's LDS-DMA alias check) and the residual-row loader.  gfx950 only.
#pragma once
#include "mfma_common.h"

namespace mv {

struct Igemm2P {
    const bf16_t* x;
    const bf16_t* w;
    const float* scale;
    const float* shift;
    const void* residual;
    void* y;
    const bf16_t* zero;
    int N, H, W, C, K, R, S, Ho, Wo, sh, sw, ph, pw, dh, dw;
    int M, tiles_m, tiles_n, act;   // M = rows covered by THIS launch (rows 0 .. M-1)
    int dbg;                        // experiments only
    long long* prof;                // experiments only: per-block phase stamps
    const bf16_t* x2;               // igemm2 DUAL: second reduction source, NHWC [N][H2][W2][C2], read at pixel stride s2
    int C2, H2, W2, s2;
    int gm;                         // igemm8: pixel tiles per group of the tile order (mfma_common.h: tile_coords)
    int skew;                       // igemm8 experiment: start delay of the first-round workgroups, (block / 8 & 3) x skew x 10 ns
    int tok;                        // > 0: head-major output y[b][n/64][t][n%64], rows m = b*tok + t (qkv projection)
    unsigned* sync;                 // igemm8s split-K: two words per tile (arrivals, partial-is-there), zero between launches; nullptr = no split
    float* ws;                      // igemm8s split-K: one fp32 tile of partial sums per tile
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int IMM> __device__ __forceinline__ void lds_read16(u32x4& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(IMM) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename OutT> struct R8;
template <> struct R8<bf16_t> {
    uint4 u;
    __device__ __forceinline__ void load(const bf16_t* p) { u = *(const uint4*)p; }
    __device__ __forceinline__ void add_to(float* v) const {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] += __uint_as_float(w[e] << 16);
            v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
        }
    }
};
template <> struct R8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *(const float4*)p; b = *(const float4*)(p + 4); }
    __device__ __forceinline__ void add_to(float* v) const {
        v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += b.x; v[5] += b.y; v[6] += b.z; v[7] += b.w;
    }
};

}  // namespace mv

// Device helpers shared by the MFMA kernels (igemm.hip, stem.hip): LDS-DMA, XCD-aware tile remap,
// vector output packing.  gfx950 only.
#pragma once
#include "common.h"

namespace mv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Lanes of ONE wave exchanging data through LDS with plain C++ stores and loads (epilogue transposes, V^T patches): the
// hardware executes a wave's DS operations in order, but the compiler may not reorder or forward them across this point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Raw buffer descriptors for branch-free bounds handling (round 5): an element outside the tensor gets the offset BUF_OOB, which is
// >= num_records, so the buffer unit drops the store / returns zeros -- no exec-masked region around the access, which is what lets
// hipcc count vmcnt exactly instead of falling back to `s_waitcnt vmcnt(0)` between the stores of an epilogue (igemm_pipe.h).
// Valid offsets must stay below 2^31: make the descriptor at the workgroup's / wave's first element, not at the tensor base.
typedef __amdgpu_buffer_rsrc_t brsrc_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
constexpr unsigned BUF_OOB = 0x80000000u;            // == num_records of every descriptor made here

__device__ __forceinline__ brsrc_t make_brsrc(const void* base, bool live = true) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, live ? (int)BUF_OOB : 0, 0x00020000);
}
__device__ __forceinline__ float4 buf_load_f4(brsrc_t r, unsigned vo, unsigned so = 0) {
    const u32x4_t u = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
    return make_float4(__uint_as_float(u[0]), __uint_as_float(u[1]), __uint_as_float(u[2]), __uint_as_float(u[3]));
}
// Stores take NO scalar offset on purpose: with an SGPR soffset hipcc's hazard recogniser does not protect the data registers of a
// > 64-bit buffer store against the next VALU write, and gfx950 does corrupt them (igemm_pipe.h: epilogue_rows).
__device__ __forceinline__ void buf_store_f4(brsrc_t r, unsigned vo, float4 v) {
    u32x4_t u;
    u[0] = __float_as_uint(v.x); u[1] = __float_as_uint(v.y); u[2] = __float_as_uint(v.z); u[3] = __float_as_uint(v.w);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, vo, 0, 0);
}
__device__ __forceinline__ uint4 buf_load_u4(brsrc_t r, unsigned vo, unsigned so = 0) {
    const u32x4_t u = __builtin_amdgcn_raw_buffer_load_b128(r, vo, so, 0);
    return make_uint4(u[0], u[1], u[2], u[3]);
}
__device__ __forceinline__ void buf_store_u4(brsrc_t r, unsigned vo, uint4 v) {
    u32x4_t u;
    u[0] = v.x; u[1] = v.y; u[2] = v.z; u[3] = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(u, r, vo, 0, 0);
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    // blocks are dispatched round-robin over the 8 XCDs; give each XCD a contiguous tile range
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// tile-list order inside an XCD: groups of GM pixel tiles; inside a group the channel tile is the
// slow index.  The ~32 blocks an XCD runs concurrently then form a (GM pixel tiles) x (32/GM channel
// tiles) patch: every x tile is shared by 32/GM blocks and every weight tile by GM blocks out of the
// XCD's 4 MB L2, instead of streaming all weight tiles through it for every pixel tile.
__device__ __forceinline__ void tile_coords(int t, int tiles_m, int tiles_n, int& tile_m, int& tile_n, int gm = 8) {
    const int GM = gm;
    const int per_group = GM * tiles_n;
    const int group = t / per_group, r = t - group * per_group;
    const int gm0 = group * GM;
    const int gsize = (tiles_m - gm0) < GM ? (tiles_m - gm0) : GM;
    tile_n = r / gsize;
    tile_m = gm0 + (r - tile_n * gsize);
}

template <typename OutT> struct Out4;
template <> struct Out4<bf16_t> {
    __device__ static __forceinline__ float4 ld(const void* p) {
        const uint2 u = *(const uint2*)p;
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u),
                           __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    __device__ static __forceinline__ void st(void* p, float4 v) {
        uint2 u;
        u.x = pack_bf2(v.x, v.y);
        u.y = pack_bf2(v.z, v.w);
        *(uint2*)p = u;
    }
};
template <> struct Out4<float> {
    __device__ static __forceinline__ float4 ld(const void* p) { return *(const float4*)p; }
    __device__ static __forceinline__ void st(void* p, float4 v) { *(float4*)p = v; }
};

template <typename OutT> struct Out8;
template <> struct Out8<bf16_t> {
    __device__ static __forceinline__ void st(bf16_t* p, const float* v) {
        uint4 u;
        u.x = pack_bf2(v[0], v[1]); u.y = pack_bf2(v[2], v[3]); u.z = pack_bf2(v[4], v[5]); u.w = pack_bf2(v[6], v[7]);
        *(uint4*)p = u;
    }
};
template <> struct Out8<float> {
    __device__ static __forceinline__ void st(float* p, const float* v) {
        *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};

// Per-lane epilogue constants for the row-major read-back phase: in every pass the lane owns channels
// n .. n+7 (n = n0 + 64*chunk + 8*(lane%8)), so scale/shift for them are loaded ONCE, unconditionally
// (clamped index), before the main loop -- never inside the per-tile epilogue, where hipcc would wrap each
// conditional load in a branch + s_waitcnt vmcnt(0) (measured: ~16 serialised L2 round trips per tile).
struct ScaleShift8 {
    float sc[8], sh[8];
    __device__ __forceinline__ void load(const float* scale, const float* shift, int n, int N) {
        const int nn = (n + 8 <= N) ? n : 0;
        float4 a = make_float4(1.f, 1.f, 1.f, 1.f), b = a, c = make_float4(0.f, 0.f, 0.f, 0.f), d = c;
        if (scale) { a = *(const float4*)(scale + nn); b = *(const float4*)(scale + nn + 4); }
        if (shift) { c = *(const float4*)(shift + nn); d = *(const float4*)(shift + nn + 4); }
        sc[0] = a.x; sc[1] = a.y; sc[2] = a.z; sc[3] = a.w; sc[4] = b.x; sc[5] = b.y; sc[6] = b.z; sc[7] = b.w;
        sh[0] = c.x; sh[1] = c.y; sh[2] = c.z; sh[3] = c.w; sh[4] = d.x; sh[5] = d.y; sh[6] = d.z; sh[7] = d.w;
    }
    __device__ __forceinline__ void apply(float* v) const {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
    }
    // Linear layers (no per-channel scale): only the 8 shift registers are ever written or read
    __device__ __forceinline__ void load_shift(const float* shift, int n, int N) {
        const int nn = (n + 8 <= N) ? n : 0;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f), d = c;
        if (shift) { c = *(const float4*)(shift + nn); d = *(const float4*)(shift + nn + 4); }
        sh[0] = c.x; sh[1] = c.y; sh[2] = c.z; sh[3] = c.w; sh[4] = d.x; sh[5] = d.y; sh[6] = d.z; sh[7] = d.w;
    }
    __device__ __forceinline__ void apply_shift(float* v) const {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += sh[e];
    }
};

}  // namespace mv

// What does one dependent kernel node of a hipGraph cost when the kernel itself does (almost) nothing?  (round 6: 41 dependent launches
// per lane in a 2.9 ms ResNet-50 step -- is the launch count worth attacking?)  A chain of N nodes on one stream, captured once,
// replayed 200 times; the same with two parallel chains (the two lanes of the forward); the same with kernels that keep every CU busy
// for ~20 us (does the gap hide behind a long kernel's tail?).   usage: ./graph_node_latency
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p[0] == 12345) p[1] = 1; }
__global__ __launch_bounds__(256) void spin(int* p, long long ticks) {           // ~ticks of the 100 MHz wall clock on every CU
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0 && p[0] == 12345) p[1] = 1;
}

static float replay(hipGraphExec_t g, hipStream_t s, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(g, s);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) hipGraphLaunch(g, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    int* buf; hipMalloc(&buf, 64); hipMemset(buf, 0, 64);
    hipStream_t s, s2; hipStreamCreate(&s); hipStreamCreate(&s2);
    for (int mode = 0; mode < 2; ++mode)                  // 0: empty kernels, 1: every CU busy for ~20 us
        for (int lanes = 1; lanes <= 2; ++lanes)
            for (int n : {1, 11, 41}) {
                hipGraph_t g; hipGraphExec_t ge;
                hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
                hipEvent_t fork, join; hipEventCreate(&fork); hipEventCreate(&join);
                if (lanes == 2) { hipEventRecord(fork, s); hipStreamWaitEvent(s2, fork, 0); }
                for (int l = 0; l < lanes; ++l)
                    for (int i = 0; i < n; ++i) {
                        hipStream_t st = l ? s2 : s;
                        if (mode == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, buf);
                        else hipLaunchKernelGGL(spin, dim3(128), dim3(256), 0, st, buf, 2000LL);
                    }
                if (lanes == 2) { hipEventRecord(join, s2); hipStreamWaitEvent(s, join, 0); }
                hipStreamEndCapture(s, &g);
                hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                const float us = replay(ge, s, 200);
                const float body = mode ? 20.0f * n : 0.f;
                printf("%s  lanes %d  %2d nodes per lane: %8.1f us per replay  -> %6.2f us per node beyond the kernel's own %4.0f us\n",
                       mode ? "20-us kernels on 128 CUs" : "empty kernels          ", lanes, n, us, (us - body) / n, mode ? 20.0f : 0.f);
                hipGraphExecDestroy(ge); hipGraphDestroy(g);
            }
    return 0;
}

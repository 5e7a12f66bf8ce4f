// A NON-PYTHON host running a whole network through the C ABI (include/eqxvision_amd.h): AlexNet's inference forward
// (reference models/classification/alexnet.py:41-90) as a fixed sequence of mv_* calls, captured ONCE into a hipGraph with
// mv_graph_begin_capture / mv_graph_end_capture and replayed with mv_graph_launch -- what `eqx.filter_jit` is to the reference.
// The host owns every buffer (hipMalloc); the library owns no state between calls.
//
//   build:  hipcc -O2 examples/c_host/alexnet_host.cpp -Iinclude -Leqxvision_amd/csrc -leqxvision_amd \
//                 -Wl,-rpath,$PWD/eqxvision_amd/csrc -o /tmp/alexnet_host
//   run:    /tmp/alexnet_host weights.bin logits.bin
// weights.bin (written by tests/test_c_host.py from a synthetic checkpoint, already in the layouts the header asks for):
//   int32 B, int32 classes, then 17 tensors as [int64 bytes][data]: x fp32 NCHW; conv1 w OIHW bf16, b fp32; conv2..5 w KRSC bf16,
//   b fp32; fc1..3 w [out][in] bf16, b fp32.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "eqxvision_amd.h"

#define HIP(x)                                                                     \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } \
    } while (0)
#define MV(x)                                                                      \
    do {                                                                           \
        int rc_ = (x);                                                             \
        if (rc_) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, mv_last_error()); exit(3); } \
    } while (0)

static void* dev_alloc(size_t n) {
    void* p = nullptr;
    HIP(hipMalloc(&p, n));
    return p;
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s weights.bin logits.bin\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t B = 0, classes = 0;
    if (fread(&B, 4, 1, f) != 1 || fread(&classes, 4, 1, f) != 1) return 1;
    std::vector<void*> t;                                   // the 17 tensors, on the device
    for (int i = 0; i < 17; ++i) {
        int64_t n = 0;
        if (fread(&n, 8, 1, f) != 1) { fprintf(stderr, "short file (tensor %d)\n", i); return 1; }
        std::vector<char> h((size_t)n);
        if (fread(h.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short file (tensor %d)\n", i); return 1; }
        void* d = dev_alloc((size_t)n);
        HIP(hipMemcpy(d, h.data(), (size_t)n, hipMemcpyHostToDevice));
        t.push_back(d);
    }
    fclose(f);
    if (mv_abi_version() != MV_ABI_VERSION) { fprintf(stderr, "ABI version %d\n", mv_abi_version()); return 1; }

    // activations: bf16 NHWC maps, bf16 rows, fp32 logits
    auto bf = [&](size_t elems) { return dev_alloc(2 * elems); };
    void *a1 = bf((size_t)B * 55 * 55 * 64), *p1 = bf((size_t)B * 27 * 27 * 64), *a2 = bf((size_t)B * 27 * 27 * 192),
         *p2 = bf((size_t)B * 13 * 13 * 192), *a3 = bf((size_t)B * 13 * 13 * 384), *a4 = bf((size_t)B * 13 * 13 * 256),
         *a5 = bf((size_t)B * 13 * 13 * 256), *p5 = bf((size_t)B * 6 * 6 * 256), *fl = bf((size_t)B * 9216), *h1 = bf((size_t)B * 4096),
         *h2 = bf((size_t)B * 4096);
    float* logits = (float*)dev_alloc(4 * (size_t)B * classes);
    hipStream_t s;
    HIP(hipStreamCreate(&s));
    const int F32 = MV_F32, BF = MV_BF16, RELU = MV_ACT_RELU, NONE = MV_ACT_NONE;

    auto forward = [&]() {
        // features (alexnet.py:44-56): conv + bias + relu in one launch each, three max-pools
        MV(mv_conv2d_nchw_fwd(t[0], t[1], nullptr, (const float*)t[2], a1, B, 3, 224, 224, 64, 11, 11, 4, 4, 2, 2, RELU, F32, BF, 0, 0, nullptr, s));
        MV(mv_maxpool2d_nhwc_fwd(a1, p1, B, 55, 55, 64, 3, 3, 2, 2, 0, 0, BF, s));
        MV(mv_conv2d_nhwc_fwd(p1, t[3], nullptr, (const float*)t[4], nullptr, a2, B, 27, 27, 64, 192, 5, 5, 1, 1, 2, 2, 1, 1, 1, RELU, BF, BF, s));
        MV(mv_maxpool2d_nhwc_fwd(a2, p2, B, 27, 27, 192, 3, 3, 2, 2, 0, 0, BF, s));
        MV(mv_conv2d_nhwc_fwd(p2, t[5], nullptr, (const float*)t[6], nullptr, a3, B, 13, 13, 192, 384, 3, 3, 1, 1, 1, 1, 1, 1, 1, RELU, BF, BF, s));
        MV(mv_conv2d_nhwc_fwd(a3, t[7], nullptr, (const float*)t[8], nullptr, a4, B, 13, 13, 384, 256, 3, 3, 1, 1, 1, 1, 1, 1, 1, RELU, BF, BF, s));
        MV(mv_conv2d_nhwc_fwd(a4, t[9], nullptr, (const float*)t[10], nullptr, a5, B, 13, 13, 256, 256, 3, 3, 1, 1, 1, 1, 1, 1, 1, RELU, BF, BF, s));
        MV(mv_maxpool2d_nhwc_fwd(a5, p5, B, 13, 13, 256, 3, 3, 2, 2, 0, 0, BF, s));
        // AdaptiveAvgPool2d((6,6)) on a 6x6 map is the identity (alexnet.py:59); jnp.ravel of the (C,H,W) sample (alexnet.py:83)
        MV(mv_nhwc_to_nchw(p5, fl, B, 256, 6, 6, BF, BF, s));
        // classifier (alexnet.py:62-70), Dropout = identity in inference
        MV(mv_linear_fwd(fl, t[11], nullptr, (const float*)t[12], nullptr, h1, B, 4096, 9216, RELU, BF, BF, s));
        MV(mv_linear_fwd(h1, t[13], nullptr, (const float*)t[14], nullptr, h2, B, 4096, 4096, RELU, BF, BF, s));
        MV(mv_linear_fwd(h2, t[15], nullptr, (const float*)t[16], nullptr, logits, B, classes, 4096, NONE, BF, F32, s));
    };

    forward();                                              // eager once (module load, kernel attributes)
    HIP(hipStreamSynchronize(s));
    void* graph = nullptr;
    MV(mv_graph_begin_capture(s));
    forward();
    MV(mv_graph_end_capture(s, &graph));
    const int reps = 20;
    MV(mv_graph_launch(graph, s));
    HIP(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; ++i) MV(mv_graph_launch(graph, s));
    HIP(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;

    std::vector<float> out((size_t)B * classes);
    HIP(hipMemcpy(out.data(), logits, out.size() * 4, hipMemcpyDeviceToHost));
    FILE* g = fopen(argv[2], "wb");
    if (!g) { perror(argv[2]); return 1; }
    fwrite(out.data(), 4, out.size(), g);
    fclose(g);
    printf("alexnet B=%d classes=%d: 12 launches captured, %.1f us per graph replay (%.0f img/s), logits[0][0..2] = %g %g %g\n", B, classes,
           us, B / us * 1e6, out[0], out[1], out[2]);
    MV(mv_graph_destroy(graph));
    return 0;
}

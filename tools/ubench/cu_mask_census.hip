// Which CUs does a stream created with hipExtStreamCreateWithCUMask(mask) run on?  Census kernel: every block records
// (XCC_ID, SE, CU) from the hardware id registers.  usage: ./cu_mask_census   (prints the placement for a few masks)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <set>
#include <map>
#include <vector>

__global__ void census(unsigned* out, int spin) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}

static std::set<unsigned> run(const char* name, const std::vector<unsigned>& mask) {
    std::set<unsigned> used;
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", name, hipGetErrorString(e)); return used; }
    const int nb = 2048;
    unsigned* d;
    hipMalloc(&d, nb * 8);
    hipMemset(d, 0xff, nb * 8);
    hipLaunchKernelGGL(census, dim3(nb), dim3(64), 0, s, d, 2000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * nb);
    hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int i = 0; i < nb; ++i) {
        const unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        used.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
    }
    printf("%-28s:", name);
    int tot = 0;
    for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total CUs %d\n", tot);
    hipFree(d);
    hipStreamDestroy(s);
    return used;
}

int main() {
    std::vector<unsigned> all(8, 0xffffffffu);
    run("all 256 bits", all);
    std::vector<unsigned> lo(8, 0); for (int i = 0; i < 4; ++i) lo[i] = 0xffffffffu;
    auto A = run("bits 0..127", lo);
    std::vector<unsigned> hi(8, 0); for (int i = 4; i < 8; ++i) hi[i] = 0xffffffffu;
    auto B = run("bits 128..255", hi);
    { int c = 0; for (unsigned u : A) c += (int)B.count(u); printf("  overlap of the two halves: %d CUs\n", c); }
    std::vector<unsigned> even(8, 0x55555555u);
    run("even bits", even);
    std::vector<unsigned> w0(8, 0); w0[0] = 0xffffffffu;
    run("bits 0..31", w0);
    std::vector<unsigned> by8(8, 0x0f0f0f0fu);
    run("nibbles 0x0f0f..", by8);
    std::vector<unsigned> b16(8, 0x0000ffffu);
    run("low half of every word", b16);
    return 0;
}

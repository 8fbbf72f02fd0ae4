// Which CUs does bit i of a hipExtStreamCreateWithCUMask mask enable on gfx950?  One block per launch slot records
// (XCC_ID, SE, SH, CU) from the hardware id registers; printed per mask.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
#include <tuple>
__global__ void k_where(unsigned* out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // spin a little so that blocks spread over every enabled CU
    long long t0 = clock64();
    while (clock64() - t0 < 20000) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
static void probe(const char* name, const std::vector<int>& bits, int words = 8) {
    std::vector<uint32_t> m(words, 0);
    for (int b : bits) m[b >> 5] |= 1u << (b & 31);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, words, m.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
    const int B = 4096;
    unsigned* d; hipMalloc(&d, B * 8);
    hipLaunchKernelGGL(k_where, dim3(B), dim3(64), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(2 * B);
    hipMemcpy(h.data(), d, B * 8, hipMemcpyDeviceToHost);
    std::set<std::tuple<int, int, int, int>> seen;
    for (int i = 0; i < B; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xF;
        seen.insert({(int)xcc, (int)((hw >> 13) & 7), (int)((hw >> 12) & 1), (int)((hw >> 8) & 15)});
    }
    printf("%-28s -> %3zu CUs:", name, seen.size());
    int shown = 0;
    for (auto& t : seen) { if (shown++ < 12) printf(" x%d.se%d.sh%d.cu%d", std::get<0>(t), std::get<1>(t), std::get<2>(t), std::get<3>(t)); }
    printf("%s\n", seen.size() > 12 ? " ..." : "");
    hipFree(d); hipStreamDestroy(st);
}
int main() {
    probe("bit 0", {0}); probe("bit 1", {1}); probe("bit 2", {2}); probe("bit 7", {7}); probe("bit 8", {8}); probe("bit 9", {9});
    probe("bit 16", {16}); probe("bit 31", {31}); probe("bit 32", {32}); probe("bit 33", {33}); probe("bit 40", {40});
    probe("bit 64", {64}); probe("bit 128", {128}); probe("bit 255", {255});
    std::vector<int> first32; for (int i = 0; i < 32; ++i) first32.push_back(i);
    probe("bits 0..31", first32);
    std::vector<int> every8; for (int i = 0; i < 256; i += 8) every8.push_back(i);
    probe("bits 0,8,16,..", every8);
    std::vector<int> all; for (int i = 0; i < 256; ++i) all.push_back(i);
    probe("bits 0..255", all);
    probe("bits 0..31 (1 word)", first32, 1);
    std::vector<int> first64; for (int i = 0; i < 64; ++i) first64.push_back(i);
    probe("bits 0..63 (2 words)", first64, 2);
    return 0;
}

// Load-shape probe for the long-window kernel (round 6): the same frame-strided non-temporal read of 1024 frames of 1080p,
// at the OCCUPANCY the window kernel runs at (2 or 3 waves per SIMD: held down with dynamic LDS), as
//   w16   16 B per lane, one frame per load instruction (1 KB per wave and frame), U loads in flight, double-buffered
//   w8     8 B per lane, one frame per load instruction (512 B per wave and frame)
//   pair  16 B per lane, TWO frames per load instruction: lanes 0..31 read 512 B of frame f, lanes 32..63 the same 512 B of frame f + 1
//         (what a lane-pair exchange -- v_permlane32_swap -- turns into 8 pixels x 2 frames per lane)
// Prints TB/s per shape: which load shape the 8-pixels-per-lane window kernel should use.
//   hipcc -O3 --offload-arch=gfx950 -o tools/load_width_probe tools/load_width_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));

// work per loaded dword: `valu` dependent-free xor/add pairs (0 = pure read), to stand in for the kernel's arithmetic
template <int U, int VALU>
__global__ void __launch_bounds__(256) k_w16(const u4v* __restrict__ p, size_t frame_vec, int frames, int tc, unsigned* out) {
    extern __shared__ unsigned lds[];
    const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= frame_vec) return;
    const int f0 = blockIdx.y * tc, f1 = min(frames, f0 + tc);
    unsigned acc = 0;
    u4v buf[2][U];
#pragma unroll
    for (int k = 0; k < U; ++k) buf[0][k] = __builtin_nontemporal_load(&p[(size_t)(f0 + k) * frame_vec + chunk]);
    for (int f = f0; f + 2 * U <= f1; f += 2 * U) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < U; ++k) buf[h ^ 1][k] = __builtin_nontemporal_load(&p[(size_t)min(f + (h + 1) * U + k, frames - 1) * frame_vec + chunk]);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                u4v v = buf[h][k];
#pragma unroll
                for (int w = 0; w < VALU; ++w) { v.x = v.x * 3u + acc; v.y ^= v.x >> 3; v.z += v.y; v.w ^= v.z; }
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc + lds[0];
}
template <int U, int VALU>
__global__ void __launch_bounds__(256) k_w8(const u2v* __restrict__ p, size_t frame_vec8, int frames, int tc, unsigned* out) {
    extern __shared__ unsigned lds[];
    const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= frame_vec8) return;
    const int f0 = blockIdx.y * tc, f1 = min(frames, f0 + tc);
    unsigned acc = 0;
    u2v buf[2][U];
#pragma unroll
    for (int k = 0; k < U; ++k) buf[0][k] = __builtin_nontemporal_load(&p[(size_t)(f0 + k) * frame_vec8 + chunk]);
    for (int f = f0; f + 2 * U <= f1; f += 2 * U) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < U; ++k) buf[h ^ 1][k] = __builtin_nontemporal_load(&p[(size_t)min(f + (h + 1) * U + k, frames - 1) * frame_vec8 + chunk]);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                u2v v = buf[h][k];
#pragma unroll
                for (int w = 0; w < VALU; ++w) { v.x = v.x * 3u + acc; v.y ^= v.x >> 3; v.x += v.y; v.y ^= v.x; }
                acc ^= v.x ^ v.y;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc + lds[0];
}
// U load instructions = 2 U frames per burst
template <int U, int VALU>
__global__ void __launch_bounds__(256) k_pair(const u4v* __restrict__ p, size_t frame_vec, int frames, int tc, unsigned* out) {
    extern __shared__ unsigned lds[];
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const size_t chunk = (size_t)wave * 32 + (lane & 31);  // 512 B per wave and frame
    if (chunk >= frame_vec) return;
    const int hi = lane >> 5;
    const int f0 = blockIdx.y * tc, f1 = min(frames, f0 + tc);
    unsigned acc = 0;
    u4v buf[2][U];
#pragma unroll
    for (int k = 0; k < U; ++k) buf[0][k] = __builtin_nontemporal_load(&p[(size_t)(f0 + 2 * k + hi) * frame_vec + chunk]);
    for (int f = f0; f + 4 * U <= f1; f += 4 * U) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int k = 0; k < U; ++k) buf[h ^ 1][k] = __builtin_nontemporal_load(&p[(size_t)min(f + (h + 1) * 2 * U + 2 * k + hi, frames - 1) * frame_vec + chunk]);
#pragma unroll
            for (int k = 0; k < U; ++k) {
                u4v v = buf[h][k];
#pragma unroll
                for (int w = 0; w < VALU; ++w) { v.x = v.x * 3u + acc; v.y ^= v.x >> 3; v.z += v.y; v.w ^= v.z; }
                acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc + lds[0];
}

template <typename F>
static void timeit(const char* name, size_t bytes, F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < 5; ++r) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-72s %.3f ms/pass  %.2f TB/s\n", name, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    if (hipGetLastError() != hipSuccess) printf("   (launch error)\n");
}
int main() {
    const size_t frame = 1920 * 1080, frames = 1024, bytes = frame * frames;
    uint8_t* d; unsigned* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 1, bytes);
    const size_t fv = frame / 16, fv8 = frame / 8;
    const int chunks = 2, tc = (int)(frames / chunks);
    for (int occ : {2, 3}) {  // waves per SIMD = 256-thread blocks per CU: 160 KB of LDS / occ, minus a margin
        const unsigned lds = occ == 2 ? 70 * 1024 : 48 * 1024;
        char nm[160];
#define RUN(KERN, U, V, GRIDX, PTR, FV, LABEL) \
        hipFuncSetAttribute((const void*)KERN<U, V>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        snprintf(nm, sizeof nm, "%d waves/SIMD  %s U=%d valu=%d", occ, LABEL, U, V); \
        timeit(nm, bytes, [&] { KERN<U, V><<<dim3((unsigned)(GRIDX), chunks), 256, lds>>>(PTR, FV, (int)frames, tc, o); });
        RUN(k_w16, 4, 0, (fv + 255) / 256, (const u4v*)d, fv, "16 B/lane, 1 frame/load (1 KB/wave/frame)")
        RUN(k_w16, 8, 0, (fv + 255) / 256, (const u4v*)d, fv, "16 B/lane, 1 frame/load (1 KB/wave/frame)")
        RUN(k_w8, 8, 0, (fv8 + 255) / 256, (const u2v*)d, fv8, " 8 B/lane, 1 frame/load (512 B/wave/frame)")
        RUN(k_w8, 16, 0, (fv8 + 255) / 256, (const u2v*)d, fv8, " 8 B/lane, 1 frame/load (512 B/wave/frame)")
        RUN(k_pair, 4, 0, (fv * 2 + 255) / 256, (const u4v*)d, fv, "16 B/lane, 2 frames/load (512 B/wave/frame)")
        RUN(k_pair, 8, 0, (fv * 2 + 255) / 256, (const u4v*)d, fv, "16 B/lane, 2 frames/load (512 B/wave/frame)")
        RUN(k_w16, 4, 6, (fv + 255) / 256, (const u4v*)d, fv, "16 B/lane, 1 frame/load (1 KB/wave/frame)")
        RUN(k_w8, 8, 6, (fv8 + 255) / 256, (const u2v*)d, fv8, " 8 B/lane, 1 frame/load (512 B/wave/frame)")
        RUN(k_pair, 4, 6, (fv * 2 + 255) / 256, (const u4v*)d, fv, "16 B/lane, 2 frames/load (512 B/wave/frame)")
        RUN(k_pair, 8, 6, (fv * 2 + 255) / 256, (const u4v*)d, fv, "16 B/lane, 2 frames/load (512 B/wave/frame)")
    }
    return 0;
}

// Pure streaming-read probe: how fast can gfx950 read HBM with 16 B / lane loads?  (calibrates the K1 roofline)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ p, size_t n, unsigned* out, int unroll8) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + 7 * stride < n; i += 8 * stride) {
        uint4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[i + k * stride];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    for (; i < n; i += stride) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// frame-strided pattern like the window kernel: a wave reads 1 KB at the same offset of consecutive 2 MB frames
__global__ void __launch_bounds__(256) k_read_frames(const uint4* __restrict__ p, size_t frame_vec, int frames, unsigned* out) {
    const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // uint4 index inside a frame
    if (chunk >= frame_vec) return;
    unsigned acc = 0;
    for (int f = 0; f + 3 < frames; f += 4) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = p[(size_t)(f + k) * frame_vec + chunk];
#pragma unroll
        for (int k = 0; k < 4; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// the same two patterns with NON-TEMPORAL loads (round 4: the window kernel's policy since the nt bit took its launch from 0.635 to 0.71)
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_read_nt(const u4v* __restrict__ p, size_t n, unsigned* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i + 7 * stride < n; i += 8 * stride) {
        u4v v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(&p[i + k * stride]);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_read_frames_nt(const u4v* __restrict__ p, size_t frame_vec, int frames, int tc, unsigned* out) {
    const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= frame_vec) return;
    const int f0 = blockIdx.y * tc, f1 = min(frames, f0 + tc);
    unsigned acc = 0;
    for (int f = f0; f + 3 < f1; f += 4) {
        u4v v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = __builtin_nontemporal_load(&p[(size_t)(f + k) * frame_vec + chunk]);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// the window kernel's traffic without its arithmetic: the same loads (1 KB per wave per frame, 4 frames per burst, chunks of
// `tc` frames along blockIdx.y) plus its stores (one u64 per 4 lanes per frame = 128 B per wave per frame)
__global__ void __launch_bounds__(256) k_read_write_frames(const uint4* __restrict__ p, size_t frame_vec, int frames, int tc,
                                                           unsigned long long* __restrict__ bits) {
    const size_t chunk = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (chunk >= frame_vec) return;
    const int f0 = blockIdx.y * tc, f1 = min(frames, f0 + tc);
    unsigned acc = 0;
    for (int f = f0; f + 3 < f1; f += 4) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = p[(size_t)(f + k) * frame_vec + chunk];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
            if ((threadIdx.x & 3) == 0) bits[(size_t)(f + k) * (frame_vec / 4) + chunk / 4] = acc;
        }
    }
}
int main() {
    const size_t frame = 1920 * 1080, frames = 1024, bytes = frame * frames, n = bytes / 16;
    uint4* d; unsigned* o;
    hipMalloc(&d, bytes); hipMalloc(&o, 4); hipMemset(d, 1, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {2048, 4096, 8192, 16384}) {
        k_read<<<blocks, 256>>>(d, n, o, 1); hipDeviceSynchronize();
        hipEventRecord(a); for (int r = 0; r < 5; ++r) k_read<<<blocks, 256>>>(d, n, o, 1); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("grid-stride read, %5d blocks: %.3f ms/pass  %.2f TB/s\n", blocks, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    }
    const size_t fv = frame / 16;
    k_read_frames<<<(fv + 255) / 256, 256>>>(d, fv, (int)frames, o); hipDeviceSynchronize();
    hipEventRecord(a); for (int r = 0; r < 5; ++r) k_read_frames<<<(fv + 255) / 256, 256>>>(d, fv, (int)frames, o); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("frame-strided read (1 KB per wave per frame, 4 frames in flight): %.3f ms/pass  %.2f TB/s\n", ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    unsigned long long* bits; hipMalloc(&bits, frames * (fv / 4) * 8);
    for (int chunks : {1, 3, 8}) {
        const int tc = (int)((frames + chunks - 1) / chunks);
        k_read_write_frames<<<dim3((fv + 255) / 256, chunks), 256>>>(d, fv, (int)frames, tc, bits); hipDeviceSynchronize();
        hipEventRecord(a); for (int r = 0; r < 5; ++r) k_read_write_frames<<<dim3((fv + 255) / 256, chunks), 256>>>(d, fv, (int)frames, tc, bits);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("frame-strided read + bit-plane write, %d chunk(s): %.3f ms/pass  %.2f TB/s read (+ %.0f MB written)\n", chunks, ms / 5,
               bytes / (ms / 5 * 1e-3) / 1e12, frames * (fv / 4) * 8 / 1e6);
    }
    for (int blocks : {4096, 16384}) {
        k_read_nt<<<blocks, 256>>>((const u4v*)d, n, o); hipDeviceSynchronize();
        hipEventRecord(a); for (int r = 0; r < 5; ++r) k_read_nt<<<blocks, 256>>>((const u4v*)d, n, o); hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("grid-stride read, NON-TEMPORAL loads, %5d blocks: %.3f ms/pass  %.2f TB/s\n", blocks, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    }
    for (int chunks : {1, 3}) {
        const int tc = (int)((frames + chunks - 1) / chunks);
        k_read_frames_nt<<<dim3((fv + 255) / 256, chunks), 256>>>((const u4v*)d, fv, (int)frames, tc, o); hipDeviceSynchronize();
        hipEventRecord(a); for (int r = 0; r < 5; ++r) k_read_frames_nt<<<dim3((fv + 255) / 256, chunks), 256>>>((const u4v*)d, fv, (int)frames, tc, o);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        printf("frame-strided read, NON-TEMPORAL loads, %d chunk(s): %.3f ms/pass  %.2f TB/s\n", chunks, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
}

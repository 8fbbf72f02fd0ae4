// LDS atomic issue cost on gfx950: cycles per wave-instruction of ds_add_rtn_u32 / ds_add_u32 / ds_read+ds_write pairs with
// scattered addresses, for 1 and 3 waves of one workgroup (the PPHT kernel's vote pattern).  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(192) k(long long* out, int iters, int spread) {
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x;
    for (int p = tid; p < 32768; p += blockDim.x) lds[p] = 0;
    __syncthreads();
    unsigned a = (tid * 2654435761u) >> 8;
    unsigned acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            a = a * 1664525u + 1013904223u;
            const unsigned idx = spread ? (a >> 10) % 30000u : (unsigned)(tid + 192 * u);
            if (MODE == 0) acc += atomicAdd(&lds[idx], 1u);
            else if (MODE == 1) atomicAdd(&lds[idx], 1u);
            else if (MODE == 2) { const unsigned v = lds[idx]; lds[idx] = v + 1; acc += v; }
            else acc += idx;
        }
    }
    const long long t1 = clock64();
    if (tid % 64 == 0) out[blockIdx.x * 4 + tid / 64] = t1 - t0;
    if (acc == 0xFFFFFFFF) out[100] = acc;
}

// The same with the sixteen addresses of an iteration computed BEFORE the timed loop (nothing but the LDS instructions and the loop
// counter inside it): what an LDS atomic costs a wave when its address is already there.
template <int MODE>
__global__ void __launch_bounds__(1024) kpre(long long* out, int iters, int spread) {
    extern __shared__ unsigned lds[];
    const int tid = threadIdx.x;
    for (int p = tid; p < 32768; p += blockDim.x) lds[p] = 0;
    __syncthreads();
    unsigned a = (tid * 2654435761u) >> 8;
    unsigned idx[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        a = a * 1664525u + 1013904223u;
        idx[u] = spread ? (a >> 10) % 30000u : (unsigned)((tid + blockDim.x * u) % 32768);
    }
    unsigned acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) acc += atomicAdd(&lds[idx[u]], 1u);
            else if (MODE == 1) atomicAdd(&lds[idx[u]], 1u);
            else { acc += lds[idx[u]]; }
        }
        asm volatile("" ::: "memory");
    }
    const long long t1 = clock64();
    if (tid % 64 == 0) out[tid / 64] = t1 - t0;
    if (acc == 0xFFFFFFFF) out[100] = acc;
}

int main() {
    long long* d;
    hipMalloc(&d, 8 * 1024);
    const char* names[] = {"ds_add_rtn_u32 (returned value used)", "ds_add_u32 (no return)", "ds_read_b32 + ds_write_b32", "address arithmetic only"};
    for (int spread = 1; spread >= 0; --spread)
        for (int threads : {64, 192})
            for (int mode = 0; mode < 4; ++mode) {
                const int iters = 256;
                void (*fn)(long long*, int, int) = mode == 0 ? k<0> : mode == 1 ? k<1> : mode == 2 ? k<2> : k<3>;
                hipLaunchKernelGGL(fn, dim3(1), dim3(threads), 131072, 0, d, iters, spread);
                hipLaunchKernelGGL(fn, dim3(1), dim3(threads), 131072, 0, d, iters, spread);
                hipDeviceSynchronize();
                long long h[4];
                hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                printf("%-40s %s  %d waves: %.1f cycles per wave-instruction\n", names[mode], spread ? "scattered" : "conflict-free", threads / 64,
                       (double)h[0] / (iters * 16));
            }
    const char* pnames[] = {"ds_add_rtn_u32, sum of the returns", "ds_add_u32 (no return)", "ds_read_b32"};
    for (int spread = 1; spread >= 0; --spread)
        for (int threads : {64, 192, 256, 512, 768, 1024})
            for (int mode = 0; mode < 3; ++mode) {
                const int iters = 256;
                void (*fn)(long long*, int, int) = mode == 0 ? kpre<0> : mode == 1 ? kpre<1> : kpre<2>;
                hipLaunchKernelGGL(fn, dim3(1), dim3(threads), 131072, 0, d, iters, spread);
                hipLaunchKernelGGL(fn, dim3(1), dim3(threads), 131072, 0, d, iters, spread);
                hipDeviceSynchronize();
                long long h[16];
                hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
                long long mx = 0;
                for (int w = 0; w < threads / 64; ++w) mx = h[w] > mx ? h[w] : mx;
                printf("addresses ready: %-36s %s  %2d waves: %.1f cycles per wave-instruction (slowest wave), %.2f per instruction of the CU\n", pnames[mode],
                       spread ? "scattered" : "conflict-free", threads / 64, (double)mx / (iters * 16), (double)mx / (iters * 16) / (threads / 64));
            }
    return 0;
}

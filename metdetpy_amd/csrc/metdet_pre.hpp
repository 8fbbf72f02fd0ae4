// metdet_pre.hpp -- loader-side preprocessing in front of the detector:
//   Transform.exec_transform  resize(INTER_LINEAR) -> BGR2GRAY -> * mask     imgproc.py:82-101, 129-139
//   MergeFunction.max         elementwise max over exp_frame processed frames utils.py:202-204
//   MaxImgContainer           running max over a clip (max_stacker)          stacker.py:43-49, 197-213
// The OpenCV fixed-point paths are restated (PARITY UNPINNED): 11-bit separable bilinear
// ((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2) >> 2 and Y = (B*3735 + G*19235 + R*9798 + 16384) >> 15
// (the 15-bit gray coefficients of OpenCV 4.x, color.hpp RY15/GY15/BY15).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct PreGeom {
    int sh, sw, dh, dw;
};

// One thread per output pixel; the e loop is the exposure merge.
__global__ void __launch_bounds__(256) k_preprocess(const uint8_t* __restrict__ src, int exp_frame, PreGeom g,
                                                    const int* __restrict__ xi, const short* __restrict__ xw,
                                                    const int* __restrict__ yi, const short* __restrict__ yw,
                                                    const uint8_t* __restrict__ mask01, uint8_t* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.dh * g.dw) return;
    // blockIdx.y = output frame of a batch: its exp_frame source frames follow each other, its output is frame y of `out`
    src += (size_t)blockIdx.y * exp_frame * g.sh * g.sw * 3;
    out += (size_t)blockIdx.y * g.dh * g.dw;
    const int dy = p / g.dw, dx = p - dy * g.dw;
    const int x0 = xi[dx], x1 = min(x0 + 1, g.sw - 1);
    const int y0 = yi[dy], y1 = min(y0 + 1, g.sh - 1);
    const int a0 = xw[2 * dx], a1 = xw[2 * dx + 1];
    const int b0 = yw[2 * dy], b1 = yw[2 * dy + 1];
    const int m = mask01 ? mask01[p] : 1;
    const size_t frame = (size_t)g.sh * g.sw * 3;
    int best = 0;
    for (int e = 0; e < exp_frame; ++e) {
        const uint8_t* f = src + (size_t)e * frame;
        const uint8_t* r0 = f + (size_t)y0 * g.sw * 3;
        const uint8_t* r1 = f + (size_t)y1 * g.sw * 3;
        int ch[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int s0 = r0[x0 * 3 + c] * a0 + r0[x1 * 3 + c] * a1;
            const int s1 = r1[x0 * 3 + c] * a0 + r1[x1 * 3 + c] * a1;
            int v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
            ch[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        const int gray = ((ch[0] * 3735 + ch[1] * 19235 + ch[2] * 9798 + 16384) >> 15) * m;
        best = max(best, gray & 0xFF);
    }
    out[p] = (uint8_t)best;
}

// The same arithmetic with the source rows staged through LDS (round 4).  The one-thread-per-pixel kernel above fetches its twelve
// taps with twelve BYTE loads scattered over two source rows: at a 4x reduction it moved the rows it touches at 3.0-3.7 TB/s.
// Here a workgroup owns PRE_SEG consecutive output pixels of ONE output row; the two source rows it needs are contiguous byte
// spans ([xi[first] * 3, (xi[last] + 2) * 3) of rows yi[dy] and yi[dy] + 1), copied into LDS with 16-byte loads -- every byte of
// a touched row crosses the memory system once, in whole cache lines -- and the taps are then read from LDS.  Requires source
// rows of a multiple of 16 bytes and a 16-byte aligned base (sw % 16 == 0: 3840, 2560, 1920, 1280 ...); other sizes keep the
// kernel above.  Bytes the launch must move: the source rows the resize touches (2 of every 4 at a 4x reduction) + the output.
typedef unsigned pre_u4 __attribute__((ext_vector_type(4)));
#define PRE_SEG 256
#define PRE_ROWS 4  // output rows per workgroup: eight source-row spans in flight per workgroup before its one barrier
// Output frames one workgroup walks through (round 6).  The counters of the launch say its waves are PARKED 79 % of the time
// (SQ_WAIT_ANY; VALU 6.5 %, LDS 2 %: gpurun_out/r06_17): not arithmetic, not LDS -- latency.  A workgroup began with three
// dependent round trips (the column table for its span, the row table for its rows, then the rows) for ONE frame's 24 KB; with
// exp_frame = 3 the same kernel ran 25 % faster per byte.  Now the tables are read once per PRE_ZB frames of the same tile.
#ifndef PRE_ZB
#define PRE_ZB 4
#endif
__global__ void __launch_bounds__(PRE_SEG) k_preprocess_rows(const uint8_t* __restrict__ src, int exp_frame, PreGeom g,
                                                              const int* __restrict__ xi, const short* __restrict__ xw,
                                                              const int* __restrict__ yi, const short* __restrict__ yw,
                                                              const uint8_t* __restrict__ mask01, uint8_t* __restrict__ out, int span_cap,
                                                              int n_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_rows[];  // [PRE_ROWS][2][span_cap]
    const int t = threadIdx.x;
    const int dx0 = blockIdx.x * PRE_SEG, dy0 = blockIdx.y * PRE_ROWS;
    const int cnt = min(PRE_SEG, g.dw - dx0);
    const int nrows = min(PRE_ROWS, g.dh - dy0);
    const int z0 = blockIdx.z * PRE_ZB, z1 = min(n_out, z0 + PRE_ZB);
    const uint8_t* const src_all = src;
    uint8_t* const out_all = out;
    const int c_first = xi[dx0], c_last = min(xi[dx0 + cnt - 1] + 1, g.sw - 1);
    const int byte0 = (c_first * 3) & ~15;
    const int span = (((c_last + 1) * 3 - byte0) + 15) & ~15;  // (<= span_cap: the host sized it from the same table)
    const int dx = dx0 + t;
    const bool live = t < cnt;
    int o0 = 0, o1 = 0, a0 = 0, a1 = 0;
    if (live) {
        const int x0 = xi[dx], x1 = min(x0 + 1, g.sw - 1);
        o0 = x0 * 3 - byte0;
        o1 = x1 * 3 - byte0;
        a0 = xw[2 * dx];
        a1 = xw[2 * dx + 1];
    }
    const size_t frame = (size_t)g.sh * g.sw * 3, row_bytes = (size_t)g.sw * 3;
    int y0s[PRE_ROWS], b0s[PRE_ROWS], b1s[PRE_ROWS];  // the row table of this tile, once
#pragma unroll
    for (int r = 0; r < PRE_ROWS; ++r) {
        const int dy = min(dy0 + r, g.dh - 1);
        y0s[r] = yi[dy];
        b0s[r] = yw[2 * dy];
        b1s[r] = yw[2 * dy + 1];
    }
    for (int z = z0; z < z1; ++z) {
    src = src_all + (size_t)z * exp_frame * frame;
    out = out_all + (size_t)z * g.dh * g.dw;
    int best[PRE_ROWS];
#pragma unroll
    for (int r = 0; r < PRE_ROWS; ++r) best[r] = 0;
    for (int e = 0; e < exp_frame; ++e) {
        const uint8_t* f = src + (size_t)e * frame + byte0;
        if (e || z > z0) __syncthreads();
#pragma unroll
        for (int r = 0; r < PRE_ROWS; ++r) {
            if (r < nrows) {
                const int y0 = y0s[r], y1 = min(y0 + 1, g.sh - 1);
                const uint8_t* r0 = f + (size_t)y0 * row_bytes;
                const uint8_t* r1 = f + (size_t)y1 * row_bytes;
                for (int o = t * 16; o < span; o += PRE_SEG * 16) {
                    // (source rows are read once: non-temporal, like the window kernel's frame loads)
                    *(pre_u4*)(lds_rows + (size_t)(2 * r) * span_cap + o) = __builtin_nontemporal_load((const pre_u4*)(r0 + o));
                    *(pre_u4*)(lds_rows + (size_t)(2 * r + 1) * span_cap + o) = __builtin_nontemporal_load((const pre_u4*)(r1 + o));
                }
            }
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int r = 0; r < PRE_ROWS; ++r) {
                if (r < nrows) {
                    const uint8_t* l0 = lds_rows + (size_t)(2 * r) * span_cap;
                    const uint8_t* l1 = l0 + span_cap;
                    const int b0 = b0s[r], b1 = b1s[r];
                    int ch[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int s0 = l0[o0 + c] * a0 + l0[o1 + c] * a1;
                        const int s1 = l1[o0 + c] * a0 + l1[o1 + c] * a1;
                        int v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
                        ch[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
                    }
                    const int gray = (ch[0] * 3735 + ch[1] * 19235 + ch[2] * 9798 + 16384) >> 15;
                    best[r] = max(best[r], gray & 0xFF);
                }
            }
        }
    }
    if (live) {
#pragma unroll
        for (int r = 0; r < PRE_ROWS; ++r)
            if (r < nrows) {
                const size_t p = (size_t)(dy0 + r) * g.dw + dx;
                out[p] = (uint8_t)(mask01 ? best[r] * mask01[p] : best[r]);
            }
    }
    }  // z
}

// Round 6, second half: the same kernel with the NEXT stage's source rows requested before the current stage is computed.  A stage is
// one source frame of one output frame (exp_frame stages per output frame, PRE_ZB output frames per workgroup).  Above, a stage is
// load -> barrier -> compute -> barrier: the workgroup's waves sit out every load's round trip (SQ_WAIT_ANY 0.79).  Here the eight
// row spans of stage s + 1 travel into registers (IT x 16 bytes per lane and span) while stage s is computed from LDS, and are
// written to LDS behind the barrier that ends stage s: one round trip per stage is hidden behind the other stage's arithmetic.
// IT = 16-byte pieces per lane and span: 1 for spans of up to 4 KB (reductions of up to 5x), 2 up to 8 KB; longer spans keep the
// kernel above.
#ifndef PRE_AHEAD
#define PRE_AHEAD 1  // stages requested ahead (A/B builds: -DPRE_AHEAD=2: two register sets)
#endif
template <int IT>
__global__ void __launch_bounds__(PRE_SEG) k_preprocess_rows_pf(const uint8_t* __restrict__ src, int exp_frame, PreGeom g,
                                                                 const int* __restrict__ xi, const short* __restrict__ xw,
                                                                 const int* __restrict__ yi, const short* __restrict__ yw,
                                                                 const uint8_t* __restrict__ mask01, uint8_t* __restrict__ out, int span_cap,
                                                                 int n_out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_rows[];  // [PRE_ROWS][2][span_cap]
    const int t = threadIdx.x;
    const int dx0 = blockIdx.x * PRE_SEG, dy0 = blockIdx.y * PRE_ROWS;
    const int cnt = min(PRE_SEG, g.dw - dx0);
    const int nrows = min(PRE_ROWS, g.dh - dy0);
    const int z0 = blockIdx.z * PRE_ZB, z1 = min(n_out, z0 + PRE_ZB);
    const int c_first = xi[dx0], c_last = min(xi[dx0 + cnt - 1] + 1, g.sw - 1);
    const int byte0 = (c_first * 3) & ~15;
    const int span = (((c_last + 1) * 3 - byte0) + 15) & ~15;  // (<= span_cap <= IT * PRE_SEG * 16: the host checked)
    const int dx = dx0 + t;
    const bool live = t < cnt;
    int o0 = 0, o1 = 0, a0 = 0, a1 = 0;
    if (live) {
        const int x0 = xi[dx], x1 = min(x0 + 1, g.sw - 1);
        o0 = x0 * 3 - byte0;
        o1 = x1 * 3 - byte0;
        a0 = xw[2 * dx];
        a1 = xw[2 * dx + 1];
    }
    const size_t frame = (size_t)g.sh * g.sw * 3, row_bytes = (size_t)g.sw * 3;
    int b0s[PRE_ROWS], b1s[PRE_ROWS];
    unsigned roff0[PRE_ROWS], roff1[PRE_ROWS];  // byte offsets of the tile's two source rows per output row, inside a source frame (a frame is < 4 GB)
#pragma unroll
    for (int r = 0; r < PRE_ROWS; ++r) {
        const int dy = min(dy0 + r, g.dh - 1);
        const int y0 = yi[dy], y1 = min(y0 + 1, g.sh - 1);
        roff0[r] = (unsigned)((size_t)y0 * row_bytes + byte0);
        roff1[r] = (unsigned)((size_t)y1 * row_bytes + byte0);
        b0s[r] = yw[2 * dy];
        b1s[r] = yw[2 * dy + 1];
    }
    const int stages = (z1 - z0) * exp_frame;
    typedef pre_u4 RegSet[2 * PRE_ROWS][IT];
    auto request = [&](RegSet& regs, int s) {  // source frame of stage s: frame (z0 * exp_frame + s) of the input
        const uint8_t* f = src + ((size_t)z0 * exp_frame + s) * frame;
#pragma unroll
        for (int r = 0; r < PRE_ROWS; ++r) {
            if (r < nrows) {
#pragma unroll
                for (int k = 0; k < IT; ++k) {
                    const int o = (t + k * PRE_SEG) * 16;
                    if (o < span) {
                        regs[2 * r][k] = __builtin_nontemporal_load((const pre_u4*)(f + roff0[r] + o));
                        regs[2 * r + 1][k] = __builtin_nontemporal_load((const pre_u4*)(f + roff1[r] + o));
                    }
                }
            }
        }
    };
    int best[PRE_ROWS];
#pragma unroll
    for (int r = 0; r < PRE_ROWS; ++r) best[r] = 0;
    int e = 0, z = z0;
    // one stage: its rows from registers to LDS, the stage PRE_AHEAD later requested into the registers just freed, the arithmetic
    auto step = [&](RegSet& regs, int s) {
        if (s) __syncthreads();  // (stage s - 1 has been computed: its rows may be overwritten)
#pragma unroll
        for (int r = 0; r < PRE_ROWS; ++r) {
            if (r < nrows) {
#pragma unroll
                for (int k = 0; k < IT; ++k) {
                    const int o = (t + k * PRE_SEG) * 16;
                    if (o < span) {
                        *(pre_u4*)(lds_rows + (size_t)(2 * r) * span_cap + o) = regs[2 * r][k];
                        *(pre_u4*)(lds_rows + (size_t)(2 * r + 1) * span_cap + o) = regs[2 * r + 1][k];
                    }
                }
            }
        }
        __syncthreads();
        if (s + PRE_AHEAD < stages) request(regs, s + PRE_AHEAD);  // in flight while this stage (and the next) is computed
        if (live) {
#pragma unroll
            for (int r = 0; r < PRE_ROWS; ++r) {
                if (r < nrows) {
                    const uint8_t* l0 = lds_rows + (size_t)(2 * r) * span_cap;
                    const uint8_t* l1 = l0 + span_cap;
                    const int b0 = b0s[r], b1 = b1s[r];
                    int ch[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int s0 = l0[o0 + c] * a0 + l0[o1 + c] * a1;
                        const int s1 = l1[o0 + c] * a0 + l1[o1 + c] * a1;
                        int v = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
                        ch[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
                    }
                    const int gray = (ch[0] * 3735 + ch[1] * 19235 + ch[2] * 9798 + 16384) >> 15;
                    best[r] = max(best[r], gray & 0xFF);
                }
            }
        }
        if (++e == exp_frame) {  // the output frame's last source frame
            if (live) {
                uint8_t* o_z = out + (size_t)z * g.dh * g.dw;
#pragma unroll
                for (int r = 0; r < PRE_ROWS; ++r)
                    if (r < nrows) {
                        const size_t p = (size_t)(dy0 + r) * g.dw + dx;
                        o_z[p] = (uint8_t)(mask01 ? best[r] * mask01[p] : best[r]);
                    }
            }
#pragma unroll
            for (int r = 0; r < PRE_ROWS; ++r) best[r] = 0;
            e = 0;
            z += 1;
        }
    };
#if PRE_AHEAD == 2
    RegSet ra, rb;
    if (stages > 0) request(ra, 0);
    if (stages > 1) request(rb, 1);
    for (int s = 0; s < stages; s += 2) {
        step(ra, s);
        if (s + 1 < stages) step(rb, s + 1);
    }
#else
    RegSet ra;
    if (stages > 0) request(ra, 0);
    for (int s = 0; s < stages; ++s) step(ra, s);
#endif
}

// acc = max(acc, frame) over K frames of `bytes` bytes each; 16 bytes per lane (requires bytes % 16 == 0 for the
// vector body, the tail goes byte-wise).
__global__ void __launch_bounds__(256) k_max_stack(const uint8_t* __restrict__ frames, int K, size_t bytes,
                                                   uint8_t* __restrict__ out) {
    const size_t nvec = bytes / 16;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid < nvec) {
        uint4 acc = ((const uint4*)frames)[tid];
        for (int k = 1; k < K; ++k) {
            const pre_u4 vv = __builtin_nontemporal_load(&((const pre_u4*)(frames + (size_t)k * bytes))[tid]);  // (every frame byte once)
            const uint4 v = make_uint4(vv.x, vv.y, vv.z, vv.w);
            const unsigned a[4] = {acc.x, acc.y, acc.z, acc.w}, b[4] = {v.x, v.y, v.z, v.w};
            unsigned r[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // per-byte unsigned max on packed bytes via two 16-bit-lane passes
                const unsigned ae = a[j] & 0x00FF00FFu, be = b[j] & 0x00FF00FFu;
                const unsigned ao = (a[j] >> 8) & 0x00FF00FFu, bo = (b[j] >> 8) & 0x00FF00FFu;
                typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                const us2 me = __builtin_elementwise_max(__builtin_bit_cast(us2, ae), __builtin_bit_cast(us2, be));
                const us2 mo = __builtin_elementwise_max(__builtin_bit_cast(us2, ao), __builtin_bit_cast(us2, bo));
                r[j] = __builtin_bit_cast(unsigned, me) | (__builtin_bit_cast(unsigned, mo) << 8);
            }
            acc = make_uint4(r[0], r[1], r[2], r[3]);
        }
        ((uint4*)out)[tid] = acc;
    }
    if (tid < bytes % 16) {
        const size_t p = nvec * 16 + tid;
        uint8_t m = frames[p];
        for (int k = 1; k < K; ++k) m = max(m, frames[(size_t)k * bytes + p]);
        out[p] = m;
    }
}

// Per-frame pixel sums, np.sum(frame) of _rf_est_kernel (videoloader.py:796-819): the only thing exposure estimation
// looks at.  grid = (blocks per frame, K); 16 bytes per lane per step, v_sad_u8 adds four bytes at a time; one 64-bit
// atomic per block.  sums must be zero on entry.
__global__ void __launch_bounds__(256) k_frame_sums(const uint8_t* __restrict__ frames, size_t bytes,
                                                    unsigned long long* __restrict__ sums) {
    const uint8_t* f = frames + (size_t)blockIdx.y * bytes;
    const size_t nvec = bytes / 16;
    unsigned acc = 0;  // <= 255 * 16 per step: no overflow below 2^20 steps per lane
    unsigned long long total = 0;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (size_t)gridDim.x * blockDim.x) {
        const uint4 q = ((const uint4*)f)[v];
        acc = __builtin_amdgcn_sad_u8(q.x, 0u, acc);
        acc = __builtin_amdgcn_sad_u8(q.y, 0u, acc);
        acc = __builtin_amdgcn_sad_u8(q.z, 0u, acc);
        acc = __builtin_amdgcn_sad_u8(q.w, 0u, acc);
        if (acc > 0xF0000000u) { total += acc; acc = 0; }
    }
    total += acc;
    if (blockIdx.x == 0 && threadIdx.x < bytes % 16) total += f[nvec * 16 + threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = total;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&sums[blockIdx.y], part[0] + part[1] + part[2] + part[3]);
}

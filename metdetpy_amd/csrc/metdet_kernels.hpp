// metdet_kernels.hpp -- device kernels of libmetdet_hip.so (gfx950 only; wave = 64 lanes).
//
// Data layout in HBM:
//   raw frames      u8  [H][W]            one buffer per frame, reached through a per-pass pointer table
//   bit images      u64 [H][WW]           WW = ceil(W/64); bit k of word w of row y  <->  pixel (y, 64*w + k)
//                                         padding bits (x >= W) are always 0
//   Hough accum     i32 [180][numrho]     one per frame slot, all-zero between uses
//
// Pipeline of one pass over K frames (every stage embarrassingly parallel over frames except the PPHT chain):
//   noise_sums -> noise_finalize -> stack_threshold -> morph_close -> dy_alive -> dy_apply -> hough_ppht
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned __int128 u128;

#define MDH_WAVE 64
#define NUM_ANGLE 180
#define MAX_LINES 512

// ------------------------------------------------------------------------------------------------
// Parameters shared by the pixel-stage kernels.
struct PassGeom {
    int H, W, WW;       // frame size, words per bit row
    int n;              // window length
    int K;              // frames in this pass
    long long t_base;   // frames consumed before this pass (frame i of the pass has timer t = t_base + 1 + i)
    int R;              // slots in the act/alive bit rings
};

__device__ __forceinline__ int window_len(const PassGeom& g, int i) {
    long long t = g.t_base + 1 + i;
    return t < g.n ? (int)t : g.n;
}

// ------------------------------------------------------------------------------------------------
// a2: noise sums over the sub-ROI (Detector.py:81-91).  For job j (a frame whose sigma is due):
//   m  = floor(sum_k x_k / k_used) per pixel (sub-window mean, utils.py:288-292)
//   S1 = sum (x_k - m),  S2 = sum (x_k - m)^2  over the last k_used frames and all ROI pixels (exact, int64)
struct NoiseJob {
    int frame;        // index i in the pass
    int k_used;       // n, or t during warm-up
    double m_cur;     // EMA momentum for this update (utils.py:360-368), computed on the host
    double one_minus; // 1 - m_cur
};

__global__ void __launch_bounds__(256) k_noise_sums(const uint8_t* const* __restrict__ ptrtab, const NoiseJob* __restrict__ jobs,
                                                    long long* __restrict__ sums, PassGeom g, int r0, int c0, int rh, int rw) {
    const NoiseJob job = jobs[blockIdx.y];
    const int npx = rh * rw;
    long long s1 = 0, s2 = 0;
    // frames of the window of pass-frame i are table entries i .. i+n-1; the last k_used of them end at i+n-1
    const int e_last = job.frame + g.n - 1;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += gridDim.x * blockDim.x) {
        const int y = r0 + p / rw, x = c0 + p % rw;
        const size_t off = (size_t)y * g.W + x;
        int tot = 0;
        for (int k = 0; k < job.k_used; ++k) tot += ptrtab[e_last - k][off];
        const int m = (tot / job.k_used) & 0xFF;
        for (int k = 0; k < job.k_used; ++k) {
            const int d = (int)ptrtab[e_last - k][off] - m;
            s1 += d;
            s2 += d * d;
        }
    }
    // wave reduction, then one atomic pair per wave
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd((unsigned long long*)&sums[2 * blockIdx.y + 0], (unsigned long long)s1);
        atomicAdd((unsigned long long*)&sums[2 * blockIdx.y + 1], (unsigned long long)s2);
    }
}

// Running scalars of LineDetector kept on the device so a pass never waits for the host.
struct ScalarState {
    double ema_v;     // stack.snr
    double sigma;     // last sigma
    double thr_f;     // bi_threshold_float
    int thr;          // bi_threshold
    int pad;
};
struct FrameScalars {
    double thr_f, snr, sigma;
    int thr, pad;
};

// a3: EMA update (utils.py:352-358) and T = round(a*snr^2 + b) (Detector.py:227-229), one thread, frame order.
__global__ void k_noise_finalize(const NoiseJob* __restrict__ jobs, int njobs, const long long* __restrict__ sums,
                                 ScalarState* __restrict__ st, FrameScalars* __restrict__ out, int* __restrict__ thr_tab,
                                 int K, double roi_px, int adaptive, double coef_a, double coef_b) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    ScalarState s = *st;
    int j = 0;
    for (int i = 0; i < K; ++i) {
        if (j < njobs && jobs[j].frame == i) {
            const double cnt = (double)jobs[j].k_used * roi_px;
            const double mu = (double)sums[2 * j] / cnt;
            double var = (double)sums[2 * j + 1] / cnt - mu * mu;
            if (var < 0) var = 0;
            s.sigma = sqrt(var);
            s.ema_v = jobs[j].m_cur * s.ema_v + jobs[j].one_minus * s.sigma;
            ++j;
        }
        if (adaptive && s.ema_v != 0) {
            s.thr_f = coef_a * (s.ema_v * s.ema_v) + coef_b;
            s.thr = (int)rint(s.thr_f);  // Python round(): half to even
        }
        thr_tab[i] = s.thr;
        out[i].thr_f = s.thr_f;
        out[i].snr = s.ema_v;
        out[i].sigma = s.sigma;
        out[i].thr = s.thr;
    }
    *st = s;
}

// ------------------------------------------------------------------------------------------------
// a1 + a4 + a6 (pointwise part): b = (max - floor(sum/len) > T), bit-packed.
// floor(s/len) < c  <=>  s < c*len for integer c, so no division is needed:
//   b = (max > T) && (sum < (max - T) * len)
// Generic-n version: one lane per pixel, n byte loads per pixel (the "algorithmic" (n+1) reads of SURVEY 8d).
__global__ void __launch_bounds__(256) k_stack_threshold_generic(const uint8_t* const* __restrict__ ptrtab,
                                                                 const int* __restrict__ thr_tab, u64* __restrict__ bbits,
                                                                 PassGeom g) {
    const int i = blockIdx.y;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int words = g.H * g.WW;
    if (gw >= words) return;
    const int y = gw / g.WW, w = gw - y * g.WW;
    const int x = w * 64 + lane;
    const bool valid = x < g.W;
    const size_t off = (size_t)y * g.W + x;
    int mx = 0, sum = 0;
    if (valid) {
        for (int k = 0; k < g.n; ++k) {
            const int v = ptrtab[i + k][off];
            mx = max(mx, v);
            sum += v;
        }
    }
    const int T = thr_tab[i];
    const int len = window_len(g, i);
    const int c = mx > T ? mx - T : 0;
    const bool b = valid && (sum < c * len);
    const u64 m = __ballot(b);
    if (lane == 0) bbits[((size_t)i * g.H + y) * g.WW + w] = m;
}

// ------------------------------------------------------------------------------------------------
// Bit-stencil helpers on a 128-bit window: position p <-> pixel x = 64*w - 3 + p.
struct Win {
    u128 valid;  // positions with 0 <= x < W
    int p_first; // position of x = 0   (may be < 0)
    int p_end;   // position of x = W   (may be > 127)
};

__device__ __forceinline__ Win make_win(int w, int W) {
    Win q;
    q.p_first = 3 - 64 * w;
    q.p_end = W - 64 * w + 3;
    const int lo = q.p_first > 0 ? q.p_first : 0;
    const int hi = q.p_end < 128 ? q.p_end : 128;
    u128 ones = ~(u128)0;
    u128 m = hi >= 128 ? ones : (((u128)1 << hi) - 1);
    m &= ~(lo > 0 ? (((u128)1 << lo) - 1) : (u128)0);
    q.valid = m;
    return q;
}

__device__ __forceinline__ u128 load_win(const u64* __restrict__ row, int w, int WW) {
    const u64 c = row[w];
    const u64 l = w > 0 ? row[w - 1] : 0ull;
    const u64 r = w + 1 < WW ? row[w + 1] : 0ull;
    return (u128)(l >> 61) | ((u128)c << 3) | ((u128)r << 67);
}

// horizontal 3-sum of a bit row with REPLICATED image border (cv2.medianBlur): returns the 2-bit count planes
__device__ __forceinline__ void hsum3_replicate(u128 b, const Win& q, u128& h0, u128& h1) {
    u128 L = b << 1, Rr = b >> 1;
    if (q.p_first >= 0 && q.p_first < 128) L = (L & ~((u128)1 << q.p_first)) | (b & ((u128)1 << q.p_first));
    if (q.p_end >= 1 && q.p_end <= 128) {
        const u128 bit = (u128)1 << (q.p_end - 1);
        Rr = (Rr & ~bit) | (b & bit);
    }
    h0 = L ^ b ^ Rr;
    h1 = (L & b) | (L & Rr) | (b & Rr);
}

// count(3 rows of 2-bit sums) >= 5
__device__ __forceinline__ u128 count_ge5(u128 a0, u128 a1, u128 b0, u128 b1, u128 c0, u128 c1) {
    const u128 s0 = a0 ^ b0 ^ c0;
    const u128 k0 = (a0 & b0) | (a0 & c0) | (b0 & c0);
    const u128 s1 = a1 ^ b1 ^ c1;
    const u128 k1 = (a1 & b1) | (a1 & c1) | (b1 & c1);
    const u128 t1 = k0 ^ s1;
    const u128 c2 = k0 & s1;
    const u128 u2 = c2 ^ k1;
    const u128 u3 = c2 & k1;
    return u3 | (u2 & (t1 | s0));
}

__device__ __forceinline__ u128 hor_or3(u128 m) { return m | (m << 1) | (m >> 1); }
// erode input: positions outside the image count as 1 (cv2.erode ignores them)
__device__ __forceinline__ u128 hor_and3_open(u128 d, const Win& q) {
    const u128 e = d | ~q.valid;
    return e & ((e << 1) | 1) & ((e >> 1) | ((u128)1 << 127));
}

// a5 + a6 + a7: act = MORPH_CLOSE( median3(diff) > T ) computed from b = (diff > T):
//   median3(diff) > T  <=>  at least 5 of the 9 (border-replicated) neighbours have diff > T.
// One thread owns one word column and a band of RB rows and slides down it with 3-row pipelines.
#define MORPH_RB 8
__global__ void __launch_bounds__(256) k_morph_close(const u64* __restrict__ bbits, u64* __restrict__ act_ring, PassGeom g) {
    const int i = blockIdx.y;
    const int nb = (g.H + MORPH_RB - 1) / MORPH_RB;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nb * g.WW) return;
    const int band = tid / g.WW, w = tid - band * g.WW;
    const int y0 = band * MORPH_RB;
    const Win q = make_win(w, g.W);
    const u64* src = bbits + (size_t)i * g.H * g.WW;
    const long long t = g.t_base + 1 + i;
    u64* dst = act_ring + (size_t)((t - 1) % g.R) * g.H * g.WW;
    u128 hA0 = 0, hA1 = 0, hB0 = 0, hB1 = 0;
    u128 mA = 0, mB = 0;
    u128 dA = 0, dB = 0;
    const u128 ones = ~(u128)0;
    for (int yy = y0 - 3; yy <= y0 + MORPH_RB + 2; ++yy) {
        const int yc = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy);
        u128 hC0, hC1;
        hsum3_replicate(load_win(src + (size_t)yc * g.WW, w, g.WW), q, hC0, hC1);
        const int ym = yy - 1;  // row whose median is complete now
        u128 mC = count_ge5(hA0, hA1, hB0, hB1, hC0, hC1) & q.valid;
        if (ym < 0 || ym >= g.H) mC = 0;  // outside the image: never wins the dilate
        const int yd = yy - 2;
        u128 dC = (hor_or3(mA) | hor_or3(mB) | hor_or3(mC)) & q.valid;
        const bool dC_in = yd >= 0 && yd < g.H;
        u128 eC = dC_in ? hor_and3_open(dC, q) : ones;  // rows outside the image never win the erode
        const int ya = yy - 3;
        if (ya >= y0 && ya < g.H) {
            const u128 a = dA & dB & eC & q.valid;
            dst[(size_t)ya * g.WW + w] = (u64)(a >> 3);
        }
        hA0 = hB0; hA1 = hB1; hB0 = hC0; hB1 = hC1;
        mA = mB; mB = mC;
        dA = dB; dB = eC;
    }
}

// a8 (temporal part): alive = NOT( act lit in all of the last len = min(n, t) frames )
// (Detector.py:236-240: sum of the last len binaries <= (len-1)*255).
__global__ void __launch_bounds__(256) k_dy_alive(const u64* __restrict__ act_ring, u64* __restrict__ alive, PassGeom g) {
    const int i = blockIdx.y;
    const int words = g.H * g.WW;
    const int gw = blockIdx.x * blockDim.x + threadIdx.x;
    if (gw >= words) return;
    const long long t = g.t_base + 1 + i;
    const int len = window_len(g, i);
    u64 all = ~0ull;
    for (int j = 0; j < len; ++j) all &= act_ring[(size_t)((t - 1 - j) % g.R) * words + gw];
    const int w = gw % g.WW;
    const int rem = g.W - 64 * w;
    const u64 vm = rem >= 64 ? ~0ull : ((1ull << rem) - 1);
    alive[(size_t)i * words + gw] = ~all & vm;
}

// a8 (spatial part) + a9 count: dst = act & erode3x3(alive); per-row and per-frame popcounts; Hough working copy.
__global__ void __launch_bounds__(256) k_dy_apply(const u64* __restrict__ act_ring, const u64* __restrict__ alive,
                                                  u64* __restrict__ dst_bits, u64* __restrict__ work_bits,
                                                  int* __restrict__ rowcount, int* __restrict__ litcount, PassGeom g,
                                                  int use_dy) {
    const int i = blockIdx.y;
    const int words = g.H * g.WW;
    const int gw = blockIdx.x * blockDim.x + threadIdx.x;
    if (gw >= words) return;
    const int y = gw / g.WW, w = gw - y * g.WW;
    const long long t = g.t_base + 1 + i;
    u64 out = act_ring[(size_t)((t - 1) % g.R) * words + gw];
    if (use_dy && out) {
        const Win q = make_win(w, g.W);
        const u64* al = alive + (size_t)i * words;
        u128 e = ~(u128)0;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= g.H) continue;
            e &= hor_and3_open(load_win(al + (size_t)yy * g.WW, w, g.WW), q);
        }
        out &= (u64)((e & q.valid) >> 3);
    }
    dst_bits[(size_t)i * words + gw] = out;
    work_bits[(size_t)i * words + gw] = out;
    if (out) {
        const int c = __popcll(out);
        atomicAdd(&rowcount[i * g.H + y], c);
        atomicAdd(&litcount[i], c);
    }
}

// bit image -> u8 {0,255} image (mdh_get_binary)
__global__ void k_expand_bits(const u64* __restrict__ bits, uint8_t* __restrict__ out, int H, int W, int WW) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    out[p] = ((bits[(size_t)y * WW + (x >> 6)] >> (x & 63)) & 1) ? 255 : 0;
}

// ------------------------------------------------------------------------------------------------
// a12 ClassicDetector (Detector.py:257-296).  f0 = newest frame.
//   m   = NOT dilate( |f1 - f0| > T )
//   dst = dilate( (|f3 - f2| > T) AND m )          (m is 0/255, so m & f is f or 0)
// Pointwise part: two bit planes per frame.
__global__ void __launch_bounds__(256) k_classic_diff(const uint8_t* const* __restrict__ ptrtab, const int* __restrict__ thr_tab,
                                                      u64* __restrict__ planes, PassGeom g) {
    const int i = blockIdx.y;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int words = g.H * g.WW;
    if (gw >= words) return;
    const int y = gw / g.WW, w = gw - y * g.WW;
    const int x = w * 64 + lane;
    const bool valid = x < g.W;
    const size_t off = (size_t)y * g.W + x;
    const int e0 = i + g.n - 1;  // newest
    bool b1 = false, b2 = false;
    if (valid) {
        // numpy's negative ring indices wrap modulo n (Detector.py:258-261)
        const int f0 = ptrtab[e0][off], f1 = ptrtab[e0 - (1 % g.n)][off];
        const int f2 = ptrtab[e0 - (2 % g.n)][off], f3 = ptrtab[e0 - (3 % g.n)][off];
        const int T = thr_tab[i];
        b1 = abs(f1 - f0) > T;
        b2 = abs(f3 - f2) > T;
    }
    const u64 m1 = __ballot(b1), m2 = __ballot(b2);
    if (lane == 0) {
        planes[((size_t)i * 2 + 0) * words + gw] = m1;
        planes[((size_t)i * 2 + 1) * words + gw] = m2;
    }
}

__global__ void __launch_bounds__(256) k_classic_stencil(const u64* __restrict__ planes, u64* __restrict__ dst_bits,
                                                         u64* __restrict__ work_bits, int* __restrict__ rowcount,
                                                         int* __restrict__ litcount, PassGeom g) {
    const int i = blockIdx.y;
    const int words = g.H * g.WW;
    const int gw = blockIdx.x * blockDim.x + threadIdx.x;
    if (gw >= words) return;
    const int y = gw / g.WW, w = gw - y * g.WW;
    const long long t = g.t_base + 1 + i;
    u64 out = 0;
    if (t >= g.n) {  // Detector.py:263: fewer than stack_maxsize frames -> no detection
        const Win q = make_win(w, g.W);
        const u64* p1 = planes + ((size_t)i * 2 + 0) * words;
        const u64* p2 = planes + ((size_t)i * 2 + 1) * words;
        u128 acc = 0;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= g.H) continue;
            // masked second difference on row yy: needs dilate(b1) at row yy -> rows yy-1..yy+1 of b1
            u128 d1 = 0;
            for (int ey = -1; ey <= 1; ++ey) {
                const int y2 = yy + ey;
                if (y2 < 0 || y2 >= g.H) continue;
                d1 |= hor_or3(load_win(p1 + (size_t)y2 * g.WW, w, g.WW));
            }
            const u128 keep = load_win(p2 + (size_t)yy * g.WW, w, g.WW) & ~d1 & q.valid;
            acc |= hor_or3(keep);
        }
        out = (u64)((acc & q.valid) >> 3);
    }
    dst_bits[(size_t)i * words + gw] = out;
    work_bits[(size_t)i * words + gw] = out;
    if (out) {
        const int c = __popcll(out);
        atomicAdd(&rowcount[i * g.H + y], c);
        atomicAdd(&litcount[i], c);
    }
}

// ------------------------------------------------------------------------------------------------
// a10: cv2.HoughLinesP = progressive probabilistic Hough, replayed exactly:
// row-major non-zero list, cv::RNG(2^64-1) visiting order, float32 rho without FMA contraction,
// vote / un-vote, 16.16 fixed-point line walk.  ONE WAVE PER FRAME: the chain over points is serial by
// definition; the 180 angles are spread 3 per lane, the peak is a wave arg-max (first maximum wins), the
// line walk tests 64 steps per ballot.  Frames are independent, so a pass runs K of these side by side.
struct HoughParams {
    int H, W, WW, numrho, half;
    int threshold, line_length, max_gap;
    int fixed_gap;         // Classic: maxLineGap = max_gap; M3: adaptive (Detector.py:342-344)
    double mask_area;
    const u64* dst_bits;   // [K][H][WW] pristine
    u64* work_bits;        // [K][H][WW] consumed
    const int* rowcount;   // [K][H]
    const int* litcount;   // [K]
    int* accum;            // [K][180*numrho], zero on entry and on exit
    unsigned* nz_orig;     // [K][nz_cap] row-major list (kept for the clean-up)
    unsigned* nz_work;     // [K][nz_cap] working list when it does not fit in LDS
    long long nz_cap;
    int* out_lines;        // [K][MAX_LINES][4]
    int* out_nraw;         // [K]
    double* out_gap;       // [K]
    const float* trig;     // [180][2] (cos, sin) as float32 of the float64 value
};

#define HOUGH_LDS_PTS 8192

__device__ __forceinline__ u64 ld_l2(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_l2(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_l2(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int rho_bin(int x, int y, float c, float s, int half) {
    // cvRound(x*cos + y*sin): float32 mul, mul, add (not fused), round half to even
    return __float2int_rn(__fadd_rn(__fmul_rn((float)x, c), __fmul_rn((float)y, s))) + half;
}

template <bool LDS_LIST>
__device__ void ppht_run(const HoughParams& P, int i, int N, int line_gap, unsigned* list) {
    const int lane = threadIdx.x;
    const size_t words = (size_t)P.H * P.WW;
    u64* work = P.work_bits + (size_t)i * words;
    unsigned* work32 = (unsigned*)work;
    int* acc = P.accum + (size_t)i * NUM_ANGLE * P.numrho;
    float tc[3], ts[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int n = lane + 64 * q;
        tc[q] = n < NUM_ANGLE ? P.trig[2 * n] : 0.f;
        ts[q] = n < NUM_ANGLE ? P.trig[2 * n + 1] : 0.f;
    }
    u64 state = 0xFFFFFFFFFFFFFFFFull;
    int found = 0;
    for (int count = N; count > 0; --count) {
        state = (u64)(unsigned)state * 4164903690ull + (state >> 32);
        const int idx = (int)((unsigned)state % (unsigned)count);
        unsigned pt;
        if (LDS_LIST) {
            pt = list[idx];
            const unsigned last = list[count - 1];
            if (lane == 0) list[idx] = last;
        } else {
            pt = ld_l2(&list[idx]);
            const unsigned last = ld_l2(&list[count - 1]);
            if (lane == 0) st_l2(&list[idx], last);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const int x = pt & 0xFFFF, y = pt >> 16;
        if (!((ld_l2(&work[(size_t)y * P.WW + (x >> 6)]) >> (x & 63)) & 1)) continue;
        // vote: ++accum[n][r], keep the first maximum
        int bv = -1, bn = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int n = lane + 64 * q;
            if (n < NUM_ANGLE) {
                const int r = rho_bin(x, y, tc[q], ts[q], P.half);
                const int val = atomicAdd(&acc[(size_t)n * P.numrho + r], 1) + 1;
                if (val > bv) { bv = val; bn = n; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int ov = __shfl_xor(bv, o), on = __shfl_xor(bn, o);
            if (ov > bv || (ov == bv && on < bn)) { bv = ov; bn = on; }
        }
        if (bv < P.threshold) continue;
        // line through (x, y) at angle bn
        const float a = -P.trig[2 * bn + 1], b = P.trig[2 * bn];
        int x0 = x, y0 = y, dx0, dy0;
        bool xflag;
        if (fabsf(a) > fabsf(b)) {
            xflag = true;
            dx0 = a > 0 ? 1 : -1;
            dy0 = __float2int_rn(__fdiv_rn(__fmul_rn(b, 65536.f), fabsf(a)));
            y0 = (y0 << 16) + 32768;
        } else {
            xflag = false;
            dy0 = b > 0 ? 1 : -1;
            dx0 = __float2int_rn(__fdiv_rn(__fmul_rn(a, 65536.f), fabsf(b)));
            x0 = (x0 << 16) + 32768;
        }
        int end_step[2], end_x[2], end_y[2];
        for (int k = 0; k < 2; ++k) {
            const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            int gap = 0, last_lit = 0;
            bool done = false;
            for (int s0 = 0; !done; s0 += 64) {
                const int s = s0 + lane;
                const int X = x0 + dx * s, Y = y0 + dy * s;
                const int j1 = xflag ? X : X >> 16, i1 = xflag ? Y >> 16 : Y;
                const bool inside = j1 >= 0 && j1 < P.W && i1 >= 0 && i1 < P.H;
                bool lit = false;
                if (inside) lit = (ld_l2(&work[(size_t)i1 * P.WW + (j1 >> 6)]) >> (j1 & 63)) & 1;
                const u64 mi = __ballot(inside), ml = __ballot(lit);
                for (int bb = 0; bb < 64; ++bb) {
                    if (!((mi >> bb) & 1)) { done = true; break; }
                    if ((ml >> bb) & 1) {
                        gap = 0;
                        last_lit = s0 + bb;
                    } else if (++gap > line_gap) {
                        done = true;
                        break;
                    }
                }
            }
            end_step[k] = last_lit;
            const int X = x0 + dx * last_lit, Y = y0 + dy * last_lit;
            end_x[k] = xflag ? X : X >> 16;
            end_y[k] = xflag ? Y >> 16 : Y;
        }
        const bool good = abs(end_x[1] - end_x[0]) >= P.line_length || abs(end_y[1] - end_y[0]) >= P.line_length;
        for (int k = 0; k < 2; ++k) {
            const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            for (int s0 = 0; s0 <= end_step[k]; s0 += 64) {
                const int s = s0 + lane;
                bool was = false;
                if (s <= end_step[k]) {
                    const int X = x0 + dx * s, Y = y0 + dy * s;
                    const int j1 = xflag ? X : X >> 16, i1 = xflag ? Y >> 16 : Y;
                    const unsigned bit = 1u << (j1 & 31);
                    const unsigned old = atomicAnd(&work32[((size_t)i1 * P.WW + (j1 >> 6)) * 2 + ((j1 >> 5) & 1)], ~bit);
                    was = old & bit;
                }
                u64 ms = __ballot(was);
                if (good) {
                    while (ms) {
                        const int bb = __builtin_ctzll(ms);
                        ms &= ms - 1;
                        const int sb = s0 + bb;
                        const int X = x0 + dx * sb, Y = y0 + dy * sb;
                        const int j1 = xflag ? X : X >> 16, i1 = xflag ? Y >> 16 : Y;
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            const int n = lane + 64 * q;
                            if (n < NUM_ANGLE) atomicSub(&acc[(size_t)n * P.numrho + rho_bin(j1, i1, tc[q], ts[q], P.half)], 1);
                        }
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (good) {
            if (found < MAX_LINES && lane == 0) {
                int* o = P.out_lines + ((size_t)i * MAX_LINES + found) * 4;
                o[0] = end_x[0]; o[1] = end_y[0]; o[2] = end_x[1]; o[3] = end_y[1];
            }
            ++found;
        }
    }
    if (lane == 0) P.out_nraw[i] = found;
}

__global__ void __launch_bounds__(64) k_hough_ppht(HoughParams P) {
    __shared__ unsigned lds_list[HOUGH_LDS_PTS];
    const int i = blockIdx.x;
    const int lane = threadIdx.x;
    const int N = P.litcount[i];
    // maxLineGap (Detector.py:342-344) -> cvRound
    double gap = P.max_gap;
    if (!P.fixed_gap) {
        const double dst_sum = (double)N / P.mask_area * 100;
        double gg = 1 - dst_sum / 0.05;
        if (gg < 0) gg = 0;
        gap = gg * P.max_gap;
    }
    const int line_gap = (int)rint(gap);
    if (lane == 0) P.out_gap[i] = gap;
    if (N == 0) {
        if (lane == 0) P.out_nraw[i] = 0;
        return;
    }
    const size_t words = (size_t)P.H * P.WW;
    const u64* dst = P.dst_bits + (size_t)i * words;
    unsigned* nzo = P.nz_orig + (size_t)i * P.nz_cap;
    // row-major list of set pixels: wave scan over the per-row counts, rows emit their own bits
    int total = 0;
    for (int yb = 0; yb < P.H; yb += 64) {
        const int y = yb + lane;
        const int c = y < P.H ? P.rowcount[i * P.H + y] : 0;
        int incl = c;
        for (int d = 1; d < 64; d <<= 1) {
            const int tt = __shfl_up(incl, d);
            if (lane >= d) incl += tt;
        }
        if (c > 0) {
            int pos = total + incl - c;
            for (int w = 0; w < P.WW; ++w) {
                u64 word = dst[(size_t)y * P.WW + w];
                while (word) {
                    const int bpos = __builtin_ctzll(word);
                    word &= word - 1;
                    nzo[pos++] = ((unsigned)y << 16) | (unsigned)(w * 64 + bpos);
                }
            }
        }
        total += __shfl(incl, 63);
    }
    __threadfence_block();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (N <= HOUGH_LDS_PTS) {
        for (int p = lane; p < N; p += 64) lds_list[p] = ld_l2(&nzo[p]);
        __syncthreads();
        ppht_run<true>(P, i, N, line_gap, lds_list);
    } else {
        unsigned* nzw = P.nz_work + (size_t)i * P.nz_cap;
        for (int p = lane; p < N; p += 64) st_l2(&nzw[p], ld_l2(&nzo[p]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ppht_run<false>(P, i, N, line_gap, nzw);
    }
    // leave the accumulator all-zero: every cell ever touched belongs to some point of the original list
    int* acc = P.accum + (size_t)i * NUM_ANGLE * P.numrho;
    for (int p = 0; p < N; ++p) {
        const unsigned pt = ld_l2(&nzo[p]);
        const int x = pt & 0xFFFF, y = pt >> 16;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int n = lane + 64 * q;
            if (n < NUM_ANGLE)
                acc[(size_t)n * P.numrho + rho_bin(x, y, P.trig[2 * n], P.trig[2 * n + 1], P.half)] = 0;
        }
    }
}

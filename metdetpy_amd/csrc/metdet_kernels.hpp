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
#include <type_traits>
#include <utility>

typedef unsigned long long u64;
typedef unsigned __int128 u128;

// f(integral_constant<int, 0>{}) ... f(integral_constant<int, N - 1>{}): a loop whose index is a compile-time constant in the body
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

#define NUM_ANGLE 180
#define HOUGH_TIERS 5  // launches a pass's frames are routed to (k_hough_classify)
#define MAX_LINES 512
#define WORKCNT_STRIDE 32  // the per-frame counts of the morphology work list sit one 128-byte line apart: packed, the appends of 32
                           // frames met in ONE L2 channel (k_morph_select 2.1 ms per pass of 512 dense 8K frames, rocprofv3 round 2c)
#define MORPH_RB 8  // rows per band of the morphology kernel (and granularity of the "band has set bits" flags)

// ------------------------------------------------------------------------------------------------
// Parameters shared by the pixel-stage kernels.
// Everything except the window kernel is short or latency-bound (serial chains, a few waves per CU).  When such a
// kernel shares a SIMD with the window kernel's VALU-saturating waves it would get one issue slot in four and its
// critical path would stretch accordingly; at raised wave priority it keeps (almost) its stand-alone latency, and the
// window kernel only loses the few slots those waves actually use.
#define RAISE_WAVE_PRIORITY() __builtin_amdgcn_s_setprio(3)
#ifndef HOUGH_WAVE_PRIO
#define HOUGH_WAVE_PRIO 3  // A/B builds: -DHOUGH_WAVE_PRIO=0..3
#endif

struct PassGeom {
    int H, W, WW;       // frame size, words per bit row
    int n;              // window length
    int K;              // frames in this pass
    int Kcap;           // frames a pass of this context holds at most (max_batch): the stride of the per-slot [Kcap][...] tables
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

// AUX: cache policy bits of the frame loads.  0 everywhere: measured with nt for the batched passes, the window kernel that follows
// got 3 % slower (1.46 -> 1.51 ms: with the default policy part of the ROI it is about to read is still in the Infinity Cache)
template <int AUX>
__global__ void __launch_bounds__(256) k_noise_sums(const uint8_t* const* __restrict__ ptrtab, const NoiseJob* __restrict__ jobs,
                                                    long long* __restrict__ sums, PassGeom g, int r0, int c0, int rh, int rw) {
    RAISE_WAVE_PRIORITY();
    // One thread = 4 horizontally adjacent pixels (one aligned dword per frame) of one ROI row.
    const NoiseJob job = jobs[blockIdx.y];
    const int k = job.k_used;
    const int xa = c0 & ~3;                      // first aligned column
    const int nx4 = (c0 + rw - xa + 3) >> 2;     // dwords per ROI row
    long long s1 = 0, s2 = 0;
    // frames of the window of pass-frame i are table entries i .. i+n-1; the last k_used of them end at i+n-1
    const int e_last = job.frame + g.n - 1;
    const int frame_bytes = g.H * g.W;
    for (int tid = blockIdx.x * blockDim.x + threadIdx.x; tid < rh * nx4; tid += gridDim.x * blockDim.x) {
        const int row = tid / nx4, x4 = tid - row * nx4;
        const int x = xa + 4 * x4;
        const int off = (r0 + row) * g.W + x;
        // one sweep: sum x and sum x^2, then  sum (x-m) = sx - k m,  sum (x-m)^2 = sxx - 2 m sx + k m^2  (exact)
        int sx[4] = {0, 0, 0, 0}, sxx[4] = {0, 0, 0, 0};
        for (int j0 = 0; j0 < k; j0 += 8) {
            unsigned v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {  // 8 independent loads in flight; slots past k read offset "far" -> 0
                const int e = e_last - min(j0 + j, k - 1);
                const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)ptrtab[e], 0, frame_bytes, 0x00020000);
                v[j] = __builtin_amdgcn_raw_buffer_load_b32(r, j0 + j < k ? off : 0x7FFFFFF0, 0, AUX);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int pv = (v[j] >> (8 * b)) & 0xFF;
                    sx[b] += pv;
                    sxx[b] += pv * pv;
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = x + b;
            if (xx >= c0 && xx < c0 + rw) {
                const int m = (sx[b] / k) & 0xFF;
                s1 += sx[b] - k * m;
                s2 += sxx[b] - 2 * m * sx[b] + k * m * m;
            }
        }
    }
    // wave reduction, block reduction through LDS, then ONE atomic pair per block (same-address atomics serialise)
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    __shared__ long long part[8];
    if ((threadIdx.x & 63) == 0) {
        part[(threadIdx.x >> 6) * 2] = s1;
        part[(threadIdx.x >> 6) * 2 + 1] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long t1 = part[0] + part[2] + part[4] + part[6], t2 = part[1] + part[3] + part[5] + part[7];
        atomicAdd((unsigned long long*)&sums[2 * blockIdx.y + 0], (unsigned long long)t1);
        atomicAdd((unsigned long long*)&sums[2 * blockIdx.y + 1], (unsigned long long)t2);
    }
}

#ifndef MDH_WINDOW_ONLY
// Same sums with one byte load per pixel: used when rows or frame bases are not 4-byte aligned.
__global__ void __launch_bounds__(256) k_noise_sums_bytes(const uint8_t* const* __restrict__ ptrtab,
                                                          const NoiseJob* __restrict__ jobs, long long* __restrict__ sums,
                                                          PassGeom g, int r0, int c0, int rh, int rw) {
    RAISE_WAVE_PRIORITY();
    const NoiseJob job = jobs[blockIdx.y];
    const int k = job.k_used;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    long long s1 = 0, s2 = 0;
    const int e_last = job.frame + g.n - 1;
    if (tid < rh * rw) {
        const int row = tid / rw, col = tid - row * rw;
        const size_t off = (size_t)(r0 + row) * g.W + c0 + col;
        int sx = 0, sxx = 0;
        for (int j = 0; j < k; ++j) {
            const int v = ptrtab[e_last - j][off];
            sx += v;
            sxx += v * v;
        }
        const int m = (sx / k) & 0xFF;
        s1 = sx - k * m;
        s2 = sxx - 2 * m * sx + k * m * m;
    }
    for (int o = 32; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
    }
    if ((threadIdx.x & 63) == 0 && (s1 | s2)) {
        atomicAdd((unsigned long long*)&sums[2 * blockIdx.y + 0], (unsigned long long)s1);
        atomicAdd((unsigned long long*)&sums[2 * blockIdx.y + 1], (unsigned long long)s2);
    }
}

#endif  // MDH_WINDOW_ONLY
// PASS TABLES (round 6).  The batched window kernels used to spend a third of their instructions on the scalar unit: per frame a
// clamped index, a 64-bit address and a load for the frame's table entry, the same for its threshold, three moves to make a buffer
// descriptor of the pointer, T * len, the window length, the "is this frame still inside my chunk" test -- ~31 scalar instructions
// beside ~32 vector ones per wave-frame at 8 pixels per lane (SQ_INST_CYCLES_SALU 0.22 of the wave cycles at n = 25, 0.15 at n = 16:
// gpurun_out/r06_03).  Now: (1) the pointer table is PADDED with WIN_PAD copies of its last entry and the frame loads are global
// loads with a scalar base (no descriptor), so an entry is one s_load at a constant offset from a running pointer; (2)
// k_noise_finalize writes one WinRec per pass-frame -- T * len replicated into both halves and -len -- padded with WIN_PAD
// records of T = 256 (no pixel can pass), loaded the same way a burst ahead; (3) every time chunk but the last is a whole number
// of window periods, so nothing is tested per frame: the frames a chunk's last period reaches past the pass are padding that
// cannot fire.  (A/B builds: -DWIDE_TABLES=0.  The per-frame instantiations keep the old form: their table arrives by mailbox.)
#ifndef WIDE_TABLES
#define WIDE_TABLES 1
#endif
#ifndef WIDE_EVEN_BURSTS
#define WIDE_EVEN_BURSTS 1  // (A/B builds: -DWIDE_EVEN_BURSTS=0: bursts of the requested size plus a remainder, as up to round 5)
#endif
#ifndef CLASSIC4_PIN_V
#define CLASSIC4_PIN_V "v87"
#endif
#ifndef WIDE_PIN152
#define WIDE_PIN152 1  // (A/B builds: -DWIDE_PIN152=0)
#endif
#define WIN_PAD 96  // >= 2 window periods + 2 load bursts of the longest window (32): how far a chunk's loads run past its last frame
struct WinRec {
    unsigned v;  // low half: min(T, 256) * len, high half: (0 - len) & 0xFFFF (len = the frame's window length): one scalar register per frame
};
__host__ __device__ inline WinRec make_winrec(unsigned T, unsigned len) { return WinRec{(T * len) | (((0u - len) & 0xFFFFu) << 16)}; }
#ifndef MDH_WINDOW_ONLY
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

// a3: EMA update (utils.py:352-358) and T = round(a*snr^2 + b) (Detector.py:227-229).
// Only the EMA itself is a chain (v_j = m_j * v_(j-1) + (1 - m_j) * sigma_j: two multiplies and an add per due frame, in the
// reference's order, no contraction); everything around it is independent per due frame.  Round 3 ran the whole body -- global
// loads of the job and its sums, two divisions, a square root, the threshold, a 32-byte global store -- inside the one-lane loop:
// 0.5 us per due frame (0.53 ms per 16 000-frame pass at 960x540, a third of the window kernel's time).  Now, per chunk of
// NF_CHUNK due frames: (A) all threads: sigma_j and the momenta into LDS; (B) one lane: the bare chain out of LDS (the loads do
// not depend on the chain, so they are issued eight steps ahead); (C) all threads: thresholds and the job results.  The values
// are those of the sequential loop bit for bit (same operations on the same operands; timeshard.NoiseChain replays them on the host).
#define NF_CHUNK 1024
#ifndef NF_THREADS
#define NF_THREADS 1024  // (A/B builds: -DNF_THREADS=256)
#endif
#define NF_FRAMES 4096  // due frames whose frame numbers are searched in LDS when the per-frame tables are filled
#ifndef NF_VGPR_CAP
#define NF_VGPR_CAP  // (A/B builds: -DNF_THREADS=256 '-DNF_VGPR_CAP=__attribute__((amdgpu_num_vgpr(56)))': a workgroup that fits beside three window waves per SIMD)
#endif
__global__ void __launch_bounds__(NF_THREADS) NF_VGPR_CAP k_noise_finalize(const NoiseJob* __restrict__ jobs, int njobs,
                                                        const long long* __restrict__ sums, ScalarState* __restrict__ st,
                                                        ScalarState* __restrict__ jobres, FrameScalars* __restrict__ out,
                                                        int* __restrict__ thr_tab, int K, double roi_px, int adaptive,
                                                        double coef_a, double coef_b, unsigned* __restrict__ nzcount, int mute_pix,
                                                        WinRec* __restrict__ wrec, int n, long long t_base) {
    RAISE_WAVE_PRIORITY();
    // the slot's list counters ([0] lit words, [1] morphology blocks): k_pass_prepare, earlier on this stream, was their last reader
    if (nzcount && threadIdx.x < 4) nzcount[threadIdx.x] = 0;  // ([2..3]: the window kernel's count of words with set bits)
    __shared__ double s_buf[4 * NF_CHUNK];
    double *const s_sig = s_buf, *const s_m = s_buf + NF_CHUNK, *const s_om = s_buf + 2 * NF_CHUNK, *const s_ema = s_buf + 3 * NF_CHUNK;
    __shared__ double s_ema_in, s_thrf_in;  // state on entry of the chunk
    __shared__ int s_thr_in;
    const int tid = threadIdx.x;
    // jobres[0] = state on entry, jobres[j+1] = state after job j
    const ScalarState s0 = *st;
    if (tid == 0) {
        jobres[0] = s0;
        s_ema_in = s0.ema_v;
        s_thrf_in = s0.thr_f;
        s_thr_in = s0.thr;
    }
    __syncthreads();
    for (int base = 0; base < njobs; base += NF_CHUNK) {
        const int cnt = min(NF_CHUNK, njobs - base);
        for (int j = tid; j < cnt; j += blockDim.x) {  // (A)
            const NoiseJob job = jobs[base + j];
            const double c = (double)job.k_used * roi_px;
            const double mu = (double)sums[2 * (base + j)] / c;
            double var = (double)sums[2 * (base + j) + 1] / c - mu * mu;
            if (var < 0) var = 0;
            s_sig[j] = sqrt(var);
            s_m[j] = job.m_cur;
            s_om[j] = job.one_minus;
        }
        __syncthreads();
        if (tid == 0) {  // (B)
            double v = s_ema_in;
#pragma unroll 8
            for (int j = 0; j < cnt; ++j) {
                v = s_m[j] * v + s_om[j] * s_sig[j];
                s_ema[j] = v;
            }
        }
        __syncthreads();
        for (int j = tid; j < cnt; j += blockDim.x) {  // (C)
            // the threshold follows the EMA only while that is non-zero (Detector.py:227: `if self.stack.snr`); otherwise it stays
            // what the last job with a non-zero EMA -- or the state on entry -- made it
            int k = j;
            while (k >= 0 && !(adaptive && s_ema[k] != 0)) --k;
            ScalarState r;
            r.ema_v = s_ema[j];
            r.sigma = s_sig[j];
            if (k >= 0) {
                r.thr_f = coef_a * (s_ema[k] * s_ema[k]) + coef_b;
                r.thr = (int)rint(r.thr_f);  // Python round(): half to even
            } else {
                r.thr_f = s_thrf_in;
                r.thr = s_thr_in;
            }
            r.pad = 0;
            jobres[base + j + 1] = r;
            if (j == cnt - 1 && base + cnt == njobs) *st = r;
        }
        __syncthreads();
        if (tid == 0 && base + cnt < njobs) {  // entry state of the next chunk (this thread wrote job 0's lane ... reads its own LDS)
            int k = cnt - 1;
            while (k >= 0 && !(adaptive && s_ema[k] != 0)) --k;
            if (k >= 0) {
                s_thrf_in = coef_a * (s_ema[k] * s_ema[k]) + coef_b;
                s_thr_in = (int)rint(s_thrf_in);
            }
            s_ema_in = s_ema[cnt - 1];
        }
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // per-frame tables: frame i takes the state after the last job whose frame is <= i (jobs are sorted by frame).  The search
    // runs on a copy of the jobs' frame numbers in LDS (ten dependent global loads per frame otherwise: the whole block waits
    // on L2 latency -- 0.1 ms per 16 000-frame pass)
    int* s_frame = (int*)s_buf;  // (the chain's arrays are free again: NF_FRAMES ints are half of them)
    const bool in_lds = njobs <= NF_FRAMES;
    if (in_lds)
        for (int j = tid; j < njobs; j += blockDim.x) s_frame[j] = jobs[j].frame;
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        int lo = 0, hi = njobs;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((in_lds ? s_frame[mid] : jobs[mid].frame) <= i) lo = mid + 1; else hi = mid;
        }
        const ScalarState s = jobres[lo];
        // mute_pix: the first frames of the pass whose windows reach in front of a seek point (mdh_seek emptied the window: they
        // would see n - 1 zero frames, i.e. a difference image that is lit almost everywhere -- megabytes of binaries for the
        // morphology and the dynamic mask to chew through, 5 ms per time window, for results that are dropped as warm-up): the
        // window kernel gets a threshold no pixel can pass (T >= 256), the reported scalars stay the chain's
        thr_tab[i] = i < mute_pix ? 256 : s.thr;
        out[i].thr_f = s.thr_f;
        out[i].snr = s.ema_v;
        out[i].sigma = s.sigma;
        out[i].thr = s.thr;
        if (wrec != nullptr) {  // the window kernel's per-frame record (see WinRec): T >= 256 can never fire, the clamp keeps T * len in 16 bits
            const long long t = t_base + 1 + i;
            const unsigned len = (unsigned)(t < n ? t : n);
            const unsigned T = (unsigned)min(max(i < mute_pix ? 256 : s.thr, 0), 256);
            wrec[i] = make_winrec(T, len);
        }
    }
    if (wrec != nullptr)  // padding behind the pass: frames that cannot fire
        for (int i = threadIdx.x; i < WIN_PAD; i += blockDim.x) wrec[K + i] = make_winrec(256u, (unsigned)n);
}

#endif  // MDH_WINDOW_ONLY
// Bit images that are almost entirely zero are kept SPARSE and TAGGED.
// (1) The thresholded difference b of a pass ([K][H][WW], per slot): the window kernel stores only the words with set bits --
// writing the zeros too made a quarter of its run time (mixed read / write HBM traffic, tools/hbm_read_probe.hip).
// (2) The ring of closed binaries (`act`, one row per frame, R rows): k_morph_close stores only the words that
// have lit pixels, each with the number of the frame it belongs to, and never clears anything.  A reader asks for
// frame t and treats a word whose tag differs (never written, or left over from frame t - k*R) as zero.
struct __attribute__((aligned(16))) ActWord {
    u64 bits;
    u64 tag;  // frame number t >= 1 (0 = never written)
};
__device__ __forceinline__ u64 act_word(const ActWord* __restrict__ row, int gw, u64 t) {
    const ActWord a = row[gw];
    return a.tag == t ? a.bits : 0ull;
}
// Block bit counts of b (sparse format) and the BITMAP of touched blocks.
// blockcnt is one per-slot allocation  [Kcap][nb * WW] counts | [Kcap][ceil(nb * WW / 32)] bitmap: bit (blk & 31) of dword blk >> 5
// of frame i says that block blk = band * WW + w has a non-zero count.  The counts and the bitmap are all-zero between passes; a
// word of b with set bits adds its population to its block and sets the block's bit.  The morphology's work selection
// (k_morph_select_touched) walks the 3x3 neighbourhoods of the set bits instead of every block of every frame, and the slot's
// next k_pass_prepare zeroes exactly the touched counts again -- round 3 zeroed (65 MB per 4000-frame 1080p pass) and scanned
// (k_morph_select: a thread per block, 16 M threads per pass) the whole table for the < 1 % of it that a quiet sky touches; the
// bitmap of such a pass is 2 MB.
// Both atomics are NON-RETURNING on purpose.  The first version of this (lists of touched block ids, appended by the add that
// found its count at zero) needed the old count and a list position back: two returning atomics in the window kernel's rare path,
// each behind an s_waitcnt vmcnt(0) that also drained the wave's eight prefetched frame loads -- 1 % of the wave-frames, and the
// launch 12-15 % slower (1080p: 1.58 -> 1.79 ms, 4K: 1.55 -> 1.85 ms on one box, gpurun_out/r04_09-10).  The bitmap is indexed by
// the block number itself (no padding per band): the rare path's address arithmetic is the peak of the window kernel's register
// use, and three registers more there (154 instead of 151 at n = 8: 160 allocated) leave no room for a Hough wave beside three
// window waves on a SIMD.
__device__ __forceinline__ unsigned block_bitmap_stride(unsigned per_frame) { return (per_frame + 31u) >> 5; }  // dwords per frame
template <bool LIST>
__device__ __forceinline__ void note_block(unsigned* __restrict__ blockcnt, const PassGeom& g, int i, int gw, unsigned pop) {
    const int nbands = (g.H + MORPH_RB - 1) / MORPH_RB;
    const unsigned per_frame = (unsigned)(nbands * g.WW);
    const unsigned blk = (unsigned)(gw / g.WW / MORPH_RB) * (unsigned)g.WW + (unsigned)(gw % g.WW);
    atomicAdd(&blockcnt[(size_t)i * per_frame + blk], pop);
    if constexpr (LIST) {
        unsigned* bitmap = blockcnt + (size_t)g.Kcap * per_frame;
        atomicOr(&bitmap[(size_t)i * block_bitmap_stride(per_frame) + (blk >> 5)], 1u << (blk & 31));
    }
}

// ------------------------------------------------------------------------------------------------
// a1 + a4 + a6 (pointwise part): b = (max - floor(sum/len) > T), bit-packed.
// floor(s/len) < c  <=>  s < c*len for integer c, so no division is needed:
//   b = (max > T) && (sum < (max - T) * len)
// Generic-n version: one lane per pixel, n byte loads per pixel (the "algorithmic" (n+1) reads of SURVEY 8d).
#ifndef MDH_WINDOW_ONLY
__global__ void __launch_bounds__(256) k_stack_threshold_generic(const uint8_t* const* __restrict__ ptrtab,
                                                                 const int* __restrict__ thr_tab, ActWord* __restrict__ bbits,
                                                                 unsigned* __restrict__ blockcnt, PassGeom g) {
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
    if (lane == 0) {
        if (m) bbits[((size_t)i * g.H + y) * g.WW + w] = ActWord{m, (u64)(g.t_base + 1 + i)};
        if (m) note_block<true>(blockcnt, g, i, gw, (unsigned)__popcll(m));
    }
}

#endif  // MDH_WINDOW_ONLY
// Time-blocked version (the fast path): one wave owns 256 pixels (4 bit words) and walks through a run of
// consecutive frames keeping the n-frame window in registers, so every frame byte is fetched ONCE per pass
// instead of n times.  Lane l holds pixels {l, 64+l, 128+l, 192+l} of the chunk as two packed u16 pairs, which
// makes the four ballots come out in natural bit order.  All arithmetic is packed 16-bit (v_pk_*_u16).
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

template <int N>
__global__ void __launch_bounds__(256) k_stack_threshold_tb(const uint8_t* const* __restrict__ ptrtab,
                                                            const int* __restrict__ thr_tab, ActWord* __restrict__ bbits,
                                                            unsigned* __restrict__ blockcnt, PassGeom g, int Tc) {
    constexpr int U = N < 8 ? N : 8;  // frames fetched per load burst
    const int lane = threadIdx.x & 63;
    const int words = g.H * g.WW;
    const int gw0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4;
    if (gw0 >= words) return;
    const int i0 = blockIdx.y * Tc;
    const int steps = min(Tc, g.K - i0);
    if (steps <= 0) return;
    // Frames are read through raw buffer descriptors (uniform base in SGPRs + 32-bit lane offset, no 64-bit
    // address math; offsets past the frame return 0, which is how padding lanes are handled).
    int off[4];
    const int frame_bytes = g.H * g.W;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int gw = gw0 + q;
        bool ok = gw < words;
        const int y = ok ? gw / g.WW : 0;
        const int w = ok ? gw - y * g.WW : 0;
        const int x = w * 64 + lane;
        ok = ok && x < g.W;
        off[q] = ok ? y * g.W + x : 0x7FFFFFFF;
    }
    auto fetch = [&](int e, unsigned& p01, unsigned& p23) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)ptrtab[e], 0, frame_bytes, 0x00020000);
        const unsigned b0 = __builtin_amdgcn_raw_buffer_load_b8(r, off[0], 0, 0);
        const unsigned b1 = __builtin_amdgcn_raw_buffer_load_b8(r, off[1], 0, 0);
        const unsigned b2 = __builtin_amdgcn_raw_buffer_load_b8(r, off[2], 0, 0);
        const unsigned b3 = __builtin_amdgcn_raw_buffer_load_b8(r, off[3], 0, 0);
        p01 = b0 | (b1 << 16);
        p23 = b2 | (b3 << 16);
    };
    union PK {
        unsigned u;
        u16x2 v;
    };
    PK w01[N], w23[N];
    u16x2 s01 = {0, 0}, s23 = {0, 0};
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
        fetch(i0 + k, w01[k].u, w23[k].u);
        s01 += w01[k].v;
        s23 += w23[k].v;
    }
    w01[N - 1].u = 0;
    w23[N - 1].u = 0;
    const int e_max = g.K + N - 2;  // last valid table entry
    for (int sb = 0; sb < steps; sb += N) {
#pragma unroll
        for (int ub = 0; ub < N; ub += U) {
            PK a01[U], a23[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (ub + u < N) {
                    const int e = min(i0 + sb + ub + u + N - 1, e_max);
                    fetch(e, a01[u].u, a23[u].u);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (ub + u < N) {
                    const int s = sb + ub + u;
                    if (s < steps) {
                        const int slot = (ub + u + N - 1) % N;
                        s01 += a01[u].v - w01[slot].v;
                        s23 += a23[u].v - w23[slot].v;
                        w01[slot] = a01[u];
                        w23[slot] = a23[u];
                        u16x2 m01 = w01[0].v, m23 = w23[0].v;
#pragma unroll
                        for (int k = 1; k < N; ++k) {
                            m01 = __builtin_elementwise_max(m01, w01[k].v);
                            m23 = __builtin_elementwise_max(m23, w23[k].v);
                        }
                        const int i = i0 + s;
                        const unsigned short T = (unsigned short)thr_tab[i];
                        const unsigned short len = (unsigned short)window_len(g, i);
                        const u16x2 Tv = {T, T}, Lv = {len, len};
                        // b = (max > T) && (sum < (max - T) * len)
                        PK e01, e23;
                        e01.v = __builtin_elementwise_sub_sat(__builtin_elementwise_sub_sat(m01, Tv) * Lv, s01);
                        e23.v = __builtin_elementwise_sub_sat(__builtin_elementwise_sub_sat(m23, Tv) * Lv, s23);
                        const u64 b0 = __ballot((e01.u & 0xFFFFu) != 0), b1 = __ballot(e01.u > 0xFFFFu);
                        const u64 b2 = __ballot((e23.u & 0xFFFFu) != 0), b3 = __ballot(e23.u > 0xFFFFu);
                        if (lane < 4 && gw0 + lane < words) {
                            const u64 v = lane == 0 ? b0 : lane == 1 ? b1 : lane == 2 ? b2 : b3;
                            if (v) bbits[(size_t)i * words + gw0 + lane] = ActWord{v, (u64)(g.t_base + 1 + i)};
                            if (v) note_block<true>(blockcnt, g, i, gw0 + lane, (unsigned)__popcll(v));  // rare
                        }
                    }
                }
            }
        }
    }
}

// Wide-load variant of the time-blocked kernel for frames whose width is a multiple of 64 (bit words are then
// contiguous runs of 64 bytes) and whose frame bases are PPL-byte aligned: a lane owns PPL = 16 (or 8) CONSECUTIVE
// pixels fetched with one 128-bit (64-bit) buffer load per frame, i.e. 16x (8x) fewer load instructions than the
// byte kernel, which is issue-bound on its byte loads.  Pixel r and pixel r + PPL/2 share a packed u16 register
// (one v_perm_b32 each), so the per-lane result bits come out as two contiguous runs.
#ifndef WIDE_OCC
#define WIDE_OCC
#endif
#ifndef WIDE_U16
#define WIDE_U16 4  // frames per load burst of the 16-pixels-per-lane kernel (A/B builds: -DWIDE_U16=8)
#endif
#ifndef WIDE_U16_BIGN
#define WIDE_U16_BIGN WIDE_U16  // the same for windows of more than 8 frames (A/B builds: -DWIDE_U16_BIGN=2: 16 registers less at n = 16)
#endif
#ifndef LEAN_AUX
#define LEAN_AUX 0  // cache policy bits of the per-frame window kernel's loads (A/B builds: 1 = sc0, 2 = nt, 16 = sc1, 17 = sc0 sc1)
#endif
#ifndef WIDE_AUX
// Cache policy bits of the batched window kernel's frame loads: nt (non-temporal).  Every frame byte is read once per launch, and
// with the default policy the CU's vector-memory front end was the limit nobody had looked at: SQ_VMEM_TA_ADDR_FIFO_FULL /
// _CMD_FIFO_FULL 15 % of the wave cycles, SQ_WAIT_INST_ANY 44 % at a VALU that is half busy (round 4, gpurun_out/r04_39, r04_41).
// Measured on one box (r04_42, isolated 1080p launch / frames per second): default 1.63-1.65 ms / 2.32-2.36 M; sc0 the same; sc1
// the same; nt 1.46-1.47 ms / 2.54-2.57 M; sc0 nt and sc1 nt like nt.  (A/B builds: -DWIDE_AUX=0 | 1 sc0 | 2 nt | 3 | 16 sc1 | 18)
#define WIDE_AUX 2
#endif
#ifndef WIDE_MAX3
#define WIDE_MAX3 1  // sliding maximum on v_pk_maximum3_f16 (A/B builds: -DWIDE_MAX3=0 keeps round-2b's three two-input maxima per frame)
#endif
#ifndef WIDE_RAW_MINN
#define WIDE_RAW_MINN 99  // window length from which the 16-pixels-per-lane kernel keeps its window as raw bytes (A/B builds: -DWIDE_RAW_MINN=2)
#endif
#ifndef WIDE_RAW8_MINN
#define WIDE_RAW8_MINN 99  // the same for the 8-pixels-per-lane kernel (A/B builds: -DWIDE_RAW8_MINN=17)
#endif
#ifndef WIDE_U8_BIGN
// frames per load burst of the 8-pixels-per-lane kernel for windows of more than 16 frames: 5 up to n = 25 (168 registers: THREE waves
// per SIMD; 1080p / n = 25 isolated 0.555 -> 0.636 of the peak against bursts of 8 at two waves, gpurun_out/r06_05), 8 above (two waves
// either way; n = 30: 0.535 against 0.514 with bursts of 5)
#define WIDE_U8_BIGN(N) ((N) <= 25 ? 5 : 8)
#endif
#ifndef WIDE_PTR_AHEAD
#define WIDE_PTR_AHEAD 1  // frame bases requested one burst ahead of the loads that use them (A/B builds: -DWIDE_PTR_AHEAD=0)
#endif
#ifndef WIDE_DENSE_CAND
#define WIDE_DENSE_CAND 1  // the dense format's candidate test inside the window kernel (see the DENSE store below; A/B builds: -DWIDE_DENSE_CAND=0)
#endif
#ifndef WIDE_RAREBITS
#define WIDE_RAREBITS 1  // gather the result bits only for waves that have one set (A/B builds: -DWIDE_RAREBITS=0)
#endif
// Three-input packed maximum.  gfx950 has no packed integer max3, but it has v_pk_maximum3_f16, and for the values this kernel
// holds (0..255 in each 16-bit half) the bit patterns are non-negative f16 denormals, whose IEEE order is their integer order;
// the kernels run with f16 denormals preserved (.amdhsa_float_denorm_mode_16_64 3, the HIP default), so the instruction returns
// exactly the integer maximum.  (tests/test_gpu_parity.py walks every window length against the oracle.)
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 pk_max3_u8(u16x2 a, u16x2 b, u16x2 c) {
    const h16x2 m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(__builtin_bit_cast(h16x2, a), __builtin_bit_cast(h16x2, b)),
                                                  __builtin_bit_cast(h16x2, c));
    return __builtin_bit_cast(u16x2, m);
}
// Per-frame protocol only (all null otherwise): see k_stack_threshold_wide.
struct LeanArgs {
    uint8_t* copy_dst;          // ring slot that receives the copy of frame 0
    u64* clr_dst;               // the slot's dst plane: the words its previous pass wrote (clr_list[0 .. clr_count[0])) are zeroed,
    const unsigned* clr_list;   //   then the list counters are reset -- what k_pass_prepare / k_noise_finalize do for a batched pass
    unsigned* clr_count;
    // "the caller's buffer has been read": the last wave to finish stores `seq` into the host word the caller polls (an event
    // record between this kernel and the next one would cost the chain ~5 us)
    unsigned* done_count;       // device counter, zero between launches
    unsigned* done_flag;        // host (pinned) word
    unsigned seq;
};
__device__ __forceinline__ void lean_wave_done(const LeanArgs& lean, unsigned total_waves) {
    if ((threadIdx.x & 63) == 0 && lean.done_count != nullptr) {
        if (atomicAdd(lean.done_count, 1u) == total_waves - 1) {
            *lean.done_count = 0;
            __hip_atomic_store(lean.done_flag, lean.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
template <int N, int PPL, bool LEAN, bool DENSE>
__device__ __forceinline__ void stack_threshold_wide_body(const uint8_t* const* __restrict__ ptrtab,
                                                              const int* __restrict__ thr_tab, ActWord* __restrict__ bbits,
                                                              unsigned* __restrict__ blockcnt, const PassGeom& g, int Tc,
                                                              const LeanArgs& lean, unsigned long long* __restrict__ nzwords,
                                                              const WinRec* __restrict__ wrec = nullptr) {
    constexpr bool TAB = WIDE_TABLES != 0 && !LEAN;  // pass tables (see WinRec)
    // DENSE (its own instantiation: the sparse one must not pay a register for it; batched passes on streams whose binaries are dense: 8K at m3det_high, where the lowered threshold lets noise
    // through and 9 % of the words carry a bit): EVERY word of b is written, plainly -- u64 [K][H][WW] in the same buffer, no
    // tags, no block counters, 128 contiguous bytes per wave and frame -- and k_morph_close_plain reads it.  The sparse-tagged
    // output that wins on a quiet sky (nothing written at all for 99 % of the wave-frames) costs such a stream a 16-byte store
    // plus a same-line atomic for four wave-frames in five (rocprofv3 PMC, round 2: 12x the bytes written, 2.5x the waiting).
    // nzwords (DENSE only): words with set bits the launch produced (one atomic per wave at the end): the host's cue to go back
    // to the sparse format (the sparse format's cue for the other direction is the pass's count of dense frames, k_morph_close).
    unsigned nz_mine = 0;  // (wave-uniform: a scalar register)
    // lean.copy_dst (per-frame protocol only, otherwise null): frame 0 of the pass is read straight from the caller's pinned
    // HOST buffer (its table entry is a host address: the loads cross PCIe) and the kernel leaves the copy the next windows
    // reach back to in the ring slot copy_dst -- no DMA command, and no DMA-to-compute hand-over, in front of the frame's
    // chain.  The first workgroup also wipes what the slot's previous frame left in dst (kept until now for get_binary / visu).
    // (LEAN is a template parameter: the batched instantiation must not pay registers for it -- 161 VGPRs / 3 waves per SIMD at
    // n = 8 and 252 / 2 at n = 16 are both at the edge of an occupancy step)
    uint8_t* __restrict__ const copy_dst = LEAN ? lean.copy_dst : nullptr;
    if (LEAN && lean.clr_count != nullptr && blockIdx.x == 0 && blockIdx.y == 0) {
        const unsigned listed = lean.clr_count[0];
        for (unsigned e = threadIdx.x; e < listed; e += 256) lean.clr_dst[lean.clr_list[e]] = 0;
        __syncthreads();
        if (threadIdx.x < 2) lean.clr_count[threadIdx.x] = 0;
    }
    constexpr int RW = PPL / 2;   // packed pair registers per window entry
    constexpr int DW = PPL / 4;   // dwords per load
    constexpr int LPW = 64 / PPL; // lanes per 64-pixel word
    constexpr int UW = N > 8 ? WIDE_U16_BIGN : WIDE_U16;
    constexpr int UREQ = PPL == 16 ? (N < UW ? N : UW) : (N < 8 ? N : (N > 16 ? WIDE_U8_BIGN(N) : 8));  // longest burst asked for
    // Bursts of (nearly) EQUAL size (round 6): NB bursts per window period, burst b = frames b N / NB .. (b + 1) N / NB - 1.  Up to
    // round 5 a period was cut into bursts of UREQ frames plus the remainder -- 8 + 8 + 8 + 1 at n = 25, 4 + 4 + 1 at n = 9: behind a
    // burst of ONE frame the wave waits for the next burst's loads with nothing left to compute, once per period.
    constexpr int NB = (N + UREQ - 1) / UREQ;                       // bursts per window period
    constexpr int U = WIDE_EVEN_BURSTS ? (N + NB - 1) / NB : UREQ;  // the longest burst (size of the burst arrays)
    typedef unsigned dvec __attribute__((ext_vector_type(DW)));
    const int lane = threadIdx.x & 63;
    const int words = g.H * g.WW;
    const int wpb = (int)(blockDim.x >> 6);  // waves per workgroup (4 by default; MDH_WIN_BLOCK: placement-granularity experiments)
    const int gw0 = (blockIdx.x * wpb + (threadIdx.x >> 6)) * PPL;  // a wave covers PPL words
    if (gw0 >= words) {
        if constexpr (LEAN) lean_wave_done(lean, gridDim.x * gridDim.y * (blockDim.x >> 6));
        return;
    }
    const int i0 = blockIdx.y * Tc;
    const int steps = min(Tc, g.K - i0);
    if (steps <= 0) {
        if constexpr (LEAN) lean_wave_done(lean, gridDim.x * gridDim.y * (blockDim.x >> 6));
        return;
    }
    const int q = lane % LPW;
    const int gw = gw0 + lane / LPW;
    const int frame_bytes = g.H * g.W;
    const int off = gw < words ? gw * 64 + q * PPL : (TAB ? 0 : 0x7FFFFFF0);  // past the frame -> zeros (TAB: any valid address; such lanes store nothing)
#if WIDE_DENSE_CAND
    // bit 0: this lane holds the first PPL pixels of an image row, bit 1: the last PPL (a set pixel on the replicated border counts twice)
    const unsigned edge_lr = DENSE ? (((gw % g.WW == 0 && q == 0) ? 1u : 0u) | ((gw % g.WW == g.WW - 1 && q == LPW - 1) ? 2u : 0u)) : 0u;
#endif
    union PK {
        unsigned u;
        u16x2 v;
    };
    typedef const __attribute__((address_space(1))) dvec* gdvec_p;
    typedef const __attribute__((address_space(1))) uint8_t* gbyte_p;
    auto fetchp = [&](const uint8_t* base) -> dvec {  // a frame's bytes of this lane, the frame's base in scalar registers
        if constexpr (TAB) {
            // global load, scalar base + 32-bit lane offset (no descriptor to assemble).  (the offset through an opaque move: extended
            // to 64 bits once outside the loop, every load would need a 64-bit vector add instead of the scalar-base address form)
            unsigned o = (unsigned)off;
            asm volatile("" : "+v"(o));
            if constexpr (WIDE_AUX == 2) return __builtin_nontemporal_load((gdvec_p)((gbyte_p)base + o));
            else return *(gdvec_p)((gbyte_p)base + o);
        }
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, frame_bytes, 0x00020000);
        if constexpr (PPL == 16) {
            return __builtin_bit_cast(dvec, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, LEAN ? LEAN_AUX : WIDE_AUX));
        } else {
            return __builtin_bit_cast(dvec, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, LEAN ? LEAN_AUX : WIDE_AUX));
        }
    };
    auto fetch = [&](int e) -> dvec {
        if constexpr (TAB) return fetchp(ptrtab[e]);
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)ptrtab[e], 0, frame_bytes, 0x00020000);
        if constexpr (PPL == 16) {
            return __builtin_bit_cast(dvec, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, LEAN ? LEAN_AUX : WIDE_AUX));
        } else {
            return __builtin_bit_cast(dvec, __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, LEAN ? LEAN_AUX : WIDE_AUX));
        }
    };
    auto unpack = [&](const dvec& d, PK (&p)[RW]) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            // low half <- byte r%4 of dword r/4 (pixel r), high half <- same byte of dword r/4 + DW/2 (pixel r + RW)
            const unsigned sel = 0x0C000C00u | (unsigned)(r & 3) | ((unsigned)(4 + (r & 3)) << 16);
            p[r].u = __builtin_amdgcn_perm(d[r / 4 + DW / 2], d[r / 4], sel);
        }
    };
    constexpr int GS = 4;                  // window slots per max group
    constexpr int NG = (N + GS - 1) / GS;  // groups
    constexpr int GL = NG - 1;             // the last group; its last slot, N-1, is where the first frame of a block goes
    constexpr int GSL = N - GL * GS;       // size of the last group
    PK win[N][RW];
    // WIDE_RAW (round 4, A/B: -DWIDE_RAW_MINN=n switches it on from window length n): the window is kept as the RAW 16 bytes per
    // lane and frame (4 registers) instead of 8 registers of unpacked pairs, and a pair is unpacked (one v_perm_b32) where it is
    // used: when its frame leaves the window (the running sum) and when its group is refilled (the sliding maximum) -- two more
    // permutes per pair and frame, half the window's registers: n = 16: 128 -> 64, n = 8: 64 -> 32 (a wave more per SIMD).
    constexpr bool MAX3_ = WIDE_MAX3 != 0 && GS == 4 && !LEAN;
    constexpr bool RAW = MAX3_ && N >= (PPL == 16 ? WIDE_RAW_MINN : WIDE_RAW8_MINN);
    dvec raw[N];
    auto W = [&](int k, int r) -> PK {
        if constexpr (RAW) {
            PK p;
            const unsigned sel = 0x0C000C00u | (unsigned)(r & 3) | ((unsigned)(4 + (r & 3)) << 16);
            p.u = __builtin_amdgcn_perm(raw[k][r / 4 + DW / 2], raw[k][r / 4], sel);
            return p;
        } else {
            return win[k][r];
        }
    };
    // Sliding maximum in 3 packed max per frame instead of (group size - 1) + (NG - 1).  The window slots are filled
    // cyclically; while group g is being refilled the maxima of the other groups do not change (oth), the part of g not
    // yet overwritten only shrinks (suffix maxima, taken once when the refill starts: R_k = max(old_{k+1}.., oth)) and the
    // part already overwritten only grows (running prefix maximum P): window max = max(P, R_k).
    u16x2 gmax[NG][RW], oth[RW], pre[RW], suf[GS > 1 ? GS - 1 : 1][RW];
    // WIDE_MAX3: with a three-input maximum the suffix maxima need not be kept: two instructions per frame instead of three and
    // one kept temporary (T = max of the group's last two old slots and the other groups) instead of three suffix registers per
    // pair (56 -> 32 registers of max state at n = 8; 161 -> 148 VGPRs at n = 8, 252 -> 236 at n = 16).  Measured: 4K / n = 16
    // 430 k -> 457 k frames/s, 1080p / n = 8 unchanged (that launch is not bound by VALU issue).  The per-frame instantiation
    // (LEAN) keeps the two-input form: its single-frame launch got 5 us slower with this one (50 -> 55 us, 8.75 k -> 8.2 k frames/s).
    constexpr bool MAX3 = WIDE_MAX3 != 0 && GS == 4 && !LEAN;
    constexpr bool RAREBITS = WIDE_RAREBITS != 0 && !LEAN && !DENSE;  // (the dense format stores every frame's word: nothing to skip)
    u16x2 t23[RW];
    // Window sums: two 16-bit fields per register, updated with plain 32-bit adds / subtracts (a field never carries or
    // borrows: 255 * (N + 1) < 2^16, and the value leaving the window is part of the sum).  v_add_u32 / v_sub_u32 issue at
    // twice the rate of the packed 16-bit forms on gfx950 (tools/valu_rate.hip).
    unsigned sum[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) sum[r] = 0;
    // the first burst of frames.  LEAN (one frame, read from the caller's buffer over PCIe): requested BEFORE the window history,
    // so that the 45 us the bus needs start at once and the history's HBM reads (n - 1 frames) run under them
    const int e_max = g.K + N - 2;
    // first frame of burst b (b = NB: the period's end) and its number of frames; static after unrolling
#define BSTART(b) (WIDE_EVEN_BURSTS ? (b) * N / NB : ((b) * U < N ? (b) * U : N))
#define BCNT(b) (BSTART((b) + 1) - BSTART(b))
    dvec buf2[2][U];
    if constexpr (LEAN) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u < BCNT(0)) buf2[0][u] = fetch(min(i0 + u + N - 1, e_max));
    }
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
        if constexpr (RAW) raw[k] = fetch(i0 + k);
        else unpack(fetch(i0 + k), win[k]);
#pragma unroll
        for (int r = 0; r < RW; ++r) sum[r] += W(k, r).u;
    }
    if constexpr (RAW) {
#pragma unroll
        for (int d = 0; d < DW; ++d) raw[N - 1][d] = 0;
    } else {
#pragma unroll
    for (int r = 0; r < RW; ++r) win[N - 1][r].u = 0;
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        // complete groups from the prologue; the state below is "about to fill slot N-1, the last slot of group GL"
#pragma unroll
        for (int gi = 0; gi < GL; ++gi) {
            u16x2 m = W(gi * GS, r).v;
#pragma unroll
            for (int k = gi * GS + 1; k < (gi + 1) * GS; ++k) m = __builtin_elementwise_max(m, W(k, r).v);
            gmax[gi][r] = m;
        }
        if constexpr (GSL > 1) {
            u16x2 m = W(GL * GS, r).v;
#pragma unroll
            for (int k = GL * GS + 1; k < N - 1; ++k) m = __builtin_elementwise_max(m, W(k, r).v);
            pre[r] = m;
            if constexpr (NG > 1) {
                u16x2 o = gmax[0][r];
#pragma unroll
                for (int h = 1; h < GL; ++h) o = __builtin_elementwise_max(o, gmax[h][r]);
                oth[r] = o;
            }
        }
    }
#ifndef WIDE_FOLD_TL
#define WIDE_FOLD_TL 1  // (round 3: only DENSE || N >= 12 -- at n = 8 it cost five registers then; with the sums taken apart as "minus the leaving frame, unpack in place, plus the new one" it costs none: 152 -> 151)
#endif
    constexpr bool FOLD_TL = WIDE_FOLD_TL;
    unsigned tl_cur = 0;  // the (T * len) * 0x10001 the sums carry at the moment (wave-uniform)
    const int len0 = window_len(g, i0);  // window_len(g, i0 + s) = min(n, len0 + s), in 32-bit scalar arithmetic
    // Bursts of U frames; the loads of burst b+1 are issued before burst b is consumed (double buffering), so a
    // wave always has loads in flight while it computes.
    // (an even number of bursts per period: the two buffers swap roles by the burst's parity, which is static after unrolling,
    // instead of being copied.  A single buffer refilled as soon as a frame is unpacked -- 16 registers less -- was measured and
    // lost 4 % stand-alone and 9 % in the pipeline at 1080p, 40 % at 4K: half the bytes in flight per wave.)
    constexpr bool PINGPONG = WIDE_MAX3 != 0 && !LEAN && (NB % 2 == 0);
    // TAB: running pointers into the padded tables (entry of the new frame of step s of the current period: pt[s], its record: wr[s])
    const uint8_t* const* pt = ptrtab + i0 + (N - 1);
    const WinRec* wr = wrec + i0;
    if constexpr (!LEAN) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u < BCNT(0)) buf2[0][u] = TAB ? fetchp(pt[u]) : fetch(min(i0 + u + N - 1, e_max));
    }
    WinRec wq2[2][U];  // TAB: the records of the burst being computed and of the next one
    if constexpr (TAB) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u < BCNT(0)) wq2[0][u] = wr[u];
    }
    // Frame bases one burst AHEAD of the loads that use them (round 4).  A load needs its frame's table entry in scalar registers;
    // fetched right in front of it, every one of a burst's U loads waited for its own scalar load (s_load, s_waitcnt, buffer_load,
    // four times over).  pq[] holds the entries of the burst after the one being computed, requested while the previous one was.
    constexpr bool PTR_AHEAD = WIDE_PTR_AHEAD != 0 && !LEAN;
    const uint8_t* pq[U];
    if constexpr (PTR_AHEAD) {
        constexpr int pp = 1 / NB, bb = 1 % NB;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u < BCNT(bb)) pq[u] = TAB ? pt[pp * N + BSTART(bb) + u] : ptrtab[min(i0 + pp * N + BSTART(bb) + u + N - 1, e_max)];
    }
    for (int sb = 0; sb < steps; sb += N) {
        // (the burst index as a compile-time constant -- static_for -- instead of an unrolled loop variable: with bursts of unequal
        // size the unroller's cost estimate no longer saw through the guards and gave up "unrolled size is too large")
        static_for<NB>([&](auto BI) __attribute__((always_inline)) {
            constexpr int bi = decltype(BI)::value;
            constexpr int ub = BSTART(bi), ucnt = BCNT(bi);
            // thresholds of the burst up front (scalar loads, off the per-frame critical path; per burst, not per window period:
            // sixteen of them at n = 16 pushed the kernel's scalar registers into spills once the frame bases travelled ahead too)
            int Tp[U];
            if constexpr (!TAB) {
#pragma unroll
            for (int u = 0; u < U; ++u)  // T >= 256 can never fire (diff <= 255); the clamp keeps T * len inside 16 bits
                if (u < ucnt) Tp[u] = min(__builtin_amdgcn_readfirstlane(thr_tab[min(i0 + sb + ub + u, g.K - 1)]), 256);
            }
            dvec(&cur)[U] = buf2[PINGPONG ? (bi & 1) : 0];
            dvec(&nxt)[U] = buf2[PINGPONG ? ((bi + 1) & 1) : 1];
            WinRec(&wcur)[U] = wq2[PINGPONG ? (bi & 1) : 0];
            WinRec(&wnxt)[U] = wq2[PINGPONG ? ((bi + 1) & 1) : 1];
            // (measured in round 2c: a workgroup barrier here, to keep the four waves on the same frames so that their loads of a
            // burst -- consecutive kilobytes of one frame -- reach the memory system together, changes nothing; eight-wave
            // workgroups with it lose 10 %)
            // prefetch the following burst (wraps into the next period)
            const int nb_first = (bi + 1 < NB) ? sb + BSTART(bi + 1) : sb + N;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in_period = u < BCNT((bi + 1) % NB);
                if (in_period) nxt[u] = PTR_AHEAD ? fetchp(pq[u]) : fetch(min(i0 + nb_first + u + N - 1, e_max));
                if constexpr (TAB) {
                    if (in_period) wnxt[u] = wr[(bi + 1 < NB) ? BSTART(bi + 1) + u : N + u];
                }
            }
            if constexpr (PTR_AHEAD) {  // table entries of the burst after that one
                const int pp = (bi + 2) / NB, bb = (bi + 2) % NB;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (u < BCNT(bb)) pq[u] = TAB ? pt[pp * N + BSTART(bb) + u] : ptrtab[min(i0 + sb + pp * N + BSTART(bb) + u + N - 1, e_max)];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u < ucnt) {
                    const int s = sb + ub + u;
                    if (TAB || s < steps) {  // (TAB: a chunk is whole periods, or runs into padding that cannot fire)
                        const int slot = (ub + u + N - 1) % N;
                        // The frame leaving the window is taken out of the sums BEFORE the new one is unpacked into its slot's
                        // registers (round 4: written as "sum + new - old" the compiler kept the old pairs alive in eight copies
                        // per frame -- v_mov, unpack in place, subtract the copy: 8 of the frame's 64 vector instructions)
                        PK a_tmp[RW];
                        if constexpr (!RAW) {
#pragma unroll
                            for (int r = 0; r < RW; ++r) sum[r] -= win[slot][r].u;
                            unpack(cur[u], win[slot]);
                        } else {
                            unpack(cur[u], a_tmp);
                        }
                        PK(&a)[RW] = RAW ? a_tmp : win[slot];
                        const int i = i0 + s;
                        if constexpr (LEAN) {
                            if (copy_dst != nullptr && i == 0 && gw < words) *(dvec*)(copy_dst + off) = cur[u];
                        }
                        // b = (max - floor(sum / len) > T)  <=>  sum + T * len - max * len < 0.  The left side is within
                        // [-255 len, 510 len], i.e. inside int16 for len <= 16 (this kernel's N), so b is the sign bit of
                        // one packed multiply-add whose addend is sum + T * len (T >= 256 can never fire; clamping it
                        // keeps T * len in 16 bits).
                        unsigned short nlen;
                        unsigned tl;
                        if constexpr (TAB) {
                            nlen = (unsigned short)(wcur[u].v >> 16);
                            tl = (wcur[u].v & 0xFFFFu) * 0x10001u;
                        } else {
                            const int len = min(g.n, len0 + s);  // window_len(g, i)
                            nlen = (unsigned short)(0 - len);
                            tl = (unsigned)__builtin_amdgcn_readfirstlane(Tp[u] * len) * 0x10001u;  // scalar unit
                        }
                        // FOLD_TL: the addend T * len lives INSIDE the running sums (round 3: one add per pixel pair and frame less): it
                        // changes with the threshold -- at most at every (interval * n)-th frame -- and with the window length
                        // during the first n frames of a stream; then every sum takes the difference (32-bit arithmetic on two
                        // 16-bit fields: exact, the fields stay below 2^16: 255 * 17 + 256 * 16).
                        if (FOLD_TL && tl != tl_cur) {
                            const unsigned delta = tl - tl_cur;
#pragma unroll
                            for (int r = 0; r < RW; ++r) sum[r] += delta;
                            tl_cur = tl;
                        }
                        const u16x2 NLv = {nlen, nlen};
                        unsigned acc = 0;
                        unsigned dk[RW];
                        // group / position of the slot the new frame goes to (all static after unrolling)
                        const int grp = slot / GS, kpos = slot % GS;
                        const int gsz = grp == GL ? GSL : GS;
#pragma unroll
                        for (int r = 0; r < RW; ++r) {
                            if constexpr (RAW) sum[r] = sum[r] + a[r].u - W(slot, r).u;
                            else sum[r] += a[r].u;
                            u16x2 m;
                            if constexpr (MAX3) {
                                const int g0 = grp * GS;
                                if (kpos == 0) {
                                    if constexpr (NG > 2) {  // maxima of the other groups, once per refill
                                        u16x2 o3[NG - 1];
                                        int c = 0;
#pragma unroll
                                        for (int h = 0; h < NG; ++h)
                                            if (h != grp) o3[c++] = gmax[h][r];
                                        u16x2 o = o3[0];
                                        int k2 = 1;
#pragma unroll
                                        for (; k2 + 1 < NG - 1; k2 += 2) o = pk_max3_u8(o, o3[k2], o3[k2 + 1]);
                                        if (k2 < NG - 1) o = __builtin_elementwise_max(o, o3[k2]);
                                        oth[r] = o;
                                    }
                                    // (no copy of the group's first new frame: the next step takes it from its window slot)
                                    if (gsz == 1) gmax[grp][r] = a[r].v;
                                } else {
                                    // running maximum of the group's new frames; the last one completes the group's maximum in place
                                    const u16x2 before = kpos == 1 ? W(g0, r).v : pre[r];
                                    if (kpos == gsz - 1) gmax[grp][r] = __builtin_elementwise_max(before, a[r].v);
                                    else pre[r] = __builtin_elementwise_max(before, a[r].v);
                                }
                                const u16x2 P = kpos == 0 ? a[r].v : (kpos == gsz - 1 ? gmax[grp][r] : pre[r]);
                                const u16x2 zero2 = {0, 0};
                                const u16x2 o = NG > 2 ? oth[r] : (NG == 2 ? gmax[grp == 0 ? 1 : 0][r] : zero2);
                                if (kpos == 0 && gsz >= 3) {
                                    // T = max(old slots 2 .. gsz-1 of the group, other groups): serves this frame and the next
                                    u16x2 T = W(g0 + 2, r).v;
                                    if (gsz == 4) T = NG > 1 ? pk_max3_u8(T, W(g0 + 3, r).v, o) : __builtin_elementwise_max(T, W(g0 + 3, r).v);
                                    else if (NG > 1) T = __builtin_elementwise_max(T, o);
                                    t23[r] = T;
                                    m = pk_max3_u8(a[r].v, W(g0 + 1, r).v, T);
                                } else if (kpos == 1 && gsz >= 3) {
                                    m = __builtin_elementwise_max(P, t23[r]);
                                } else {
                                    // at most one old slot of the group is left (kpos >= gsz - 2)
                                    const bool one_old = kpos == gsz - 2;
                                    if (one_old && NG > 1) m = pk_max3_u8(P, W(g0 + gsz - 1, r).v, o);
                                    else if (one_old) m = __builtin_elementwise_max(P, W(g0 + gsz - 1, r).v);
                                    else if (NG > 1) m = __builtin_elementwise_max(P, o);
                                    else m = P;
                                }
                            } else {
                            if (kpos == 0) {  // the refill of group grp starts: maxima of the other groups, suffix maxima of this one
                                if constexpr (NG > 1) {
                                    u16x2 o = gmax[grp == 0 ? 1 : 0][r];
#pragma unroll
                                    for (int h = 0; h < NG; ++h)
                                        if (h != grp && h != (grp == 0 ? 1 : 0)) o = __builtin_elementwise_max(o, gmax[h][r]);
                                    oth[r] = o;
                                }
                                if (gsz >= 2) {
                                    u16x2 sfx = win[grp * GS + gsz - 1][r].v;
                                    suf[gsz - 2][r] = NG > 1 ? __builtin_elementwise_max(sfx, oth[r]) : sfx;
#pragma unroll
                                    for (int kk = GS - 3; kk >= 0; --kk)
                                        if (kk <= gsz - 3) {
                                            sfx = __builtin_elementwise_max(win[grp * GS + kk + 1][r].v, sfx);
                                            suf[kk][r] = NG > 1 ? __builtin_elementwise_max(sfx, oth[r]) : sfx;
                                        }
                                }
                                pre[r] = a[r].v;
                            } else {
                                pre[r] = __builtin_elementwise_max(pre[r], a[r].v);
                            }
                            m = pre[r];
                            if (kpos < gsz - 1) m = __builtin_elementwise_max(m, suf[kpos][r]);
                            else if (NG > 1) m = __builtin_elementwise_max(m, oth[r]);
                            if (kpos == gsz - 1) gmax[grp][r] = pre[r];
                            }
                            // sum < (max - T) * len  <=>  bit 15 of sum + (max - T) * (-len) mod 2^16 (all terms < 2^15 for
                            // n <= 128); written as a multiply-add so that it maps to one v_pk_mad_u16
                            PK d, st;
                            st.u = FOLD_TL ? sum[r] : sum[r] + tl;
                            d.v = m * NLv + st.v;
                            if constexpr (RAREBITS) {
                                dk[r] = d.u;
                                acc |= d.u;
                            } else {
                                acc = (acc >> 1) | (d.u & 0x80008000u);
                            }
                        }
                        if constexpr (RAW) raw[slot] = cur[u];  // (behind the loop: the leaving frame's pairs were unpacked from it above)
                        if constexpr (RAREBITS) {
                            // a wave of 1024 pixels rarely holds a pixel over the threshold (noise at 4-5 sigma, a streak every few
                            // dozen waves): the sign bits are only gathered when the OR of the wave's results has one set
                            if (__ballot((acc & 0x80008000u) != 0) == 0) continue;
                            acc = 0;
#pragma unroll
                            for (int r = 0; r < RW; ++r) acc = (acc >> 1) | (dk[r] & 0x80008000u);
                        }
                        // reg r landed on bit 8+r-(8-RW) ... after RW shifts: low half bits (16-RW)..15 = pixels 0..RW-1,
                        // high half bits (32-RW)..31 = pixels RW..PPL-1
                        const unsigned bits = ((acc >> (16 - RW)) & ((1u << RW) - 1)) | ((acc >> (32 - 2 * RW)) & (((1u << RW) - 1) << RW));
                        u64 v;
                        if constexpr (PPL == 16) {
                            // the four lanes of a word hold 16 result bits each: gather them in lane q == 0 with two
                            // quad-permute DPP moves (lane 0 <- 1, lane 2 <- 3; then lane 0 <- 2)
                            const unsigned odd = (unsigned)__builtin_amdgcn_mov_dpp((int)bits, 0xF5, 0xF, 0xF, true);
                            const unsigned pair = bits | (odd << 16);
                            const unsigned upper = (unsigned)__builtin_amdgcn_mov_dpp((int)pair, 0xAA, 0xF, 0xF, true);
                            v = (u64)pair | ((u64)upper << 32);
                        } else {
                            v = (u64)bits << (q * PPL);
#pragma unroll
                            for (int o = 1; o < LPW; o <<= 1) v |= __shfl_xor(v, o);
                        }
                        if constexpr (DENSE) {
                            const bool in_pass = !TAB || i < g.K;  // (TAB: the last period of the last chunk may reach past the pass)
                            nz_mine += (unsigned)__popcll(__ballot(q == 0 && gw < words && v != 0 && in_pass));
                            if (q == 0 && gw < words && in_pass) __builtin_nontemporal_store(v, &((u64*)bbits)[(size_t)i * words + gw]);
#if WIDE_DENSE_CAND
                            // CANDIDATE BLOCKS (round 5).  The median keeps a pixel only with >= 5 set samples in its 3x3 window, i.e.
                            // only where SOME ROW holds two set bits at most two columns apart (on the replicated left / right image
                            // border a single bit counts twice).  On the noise a lowered threshold lets through (8K at m3det_high: 9 %
                            // of the words carry a bit, 0.03 % such a pair) that names a few hundred of a frame's 65 000 blocks, and
                            // the morphology closes those from the plain words instead of reading all 2 GB of them back (0.99 -> 0.25 ms
                            // per 500-frame pass, 118 k -> 128-135 k frames/s, 0.56 -> 0.60-0.64 of the peak at the pipeline's pace).
                            // The test runs on the lanes' own 16-bit masks BEFORE they are gathered into words -- 32-bit operations,
                            // the left lane's top two bits over one DPP wave shift: ~2 % on this launch (the first version, on the
                            // gathered 64-bit word in every lane, cost 13 %).  The wave's first lane and the first / last lane of an
                            // image row have no neighbour to ask and take their edge bits for a pair (a few false candidates, never
                            // a missed one).  Flagged through the block bitmap of the sparse format, non-returning atomics in a
                            // rare path (~1 % of the wave-frames).
                            if (blockcnt != nullptr) {
                                const unsigned lt = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(bits >> (PPL - 2)), 0x138, 0xF, 0xF, false);  // wave_shr:1
                                const unsigned pair = bits & ((bits >> 1) | (bits >> 2));
                                const unsigned lo2 = bits & 3u;
                                const unsigned cross = lane == 0 ? lo2 : ((lo2 & (0u - (lt != 0))) & 1u) | (lo2 & lt & 2u);
                                const unsigned edge = (bits & edge_lr & 1u) | ((bits >> (PPL - 1)) & (edge_lr >> 1));
                                const bool cand = (pair | cross | edge) != 0 && gw < words && in_pass;
                                if (__ballot(cand) != 0) {
                                    if (cand) {
                                        // the blocks this pair can light up: what comes out of median + close lies within 2 rows and
                                        // 2 columns of the pair's bits -- its own block, the block above / below when the row is one of
                                        // the block's first / last two, the block to the left / right when the bits sit at the word's
                                        // edge (this lane's bits 0..2 in the word's first lane, 13..15 in its last), and the diagonal
                                        // ones when both hold.  The selection then lists exactly the flagged blocks (no dilation).
                                        int gwr = gw;
                                        asm volatile("" : "+v"(gwr));
                                        const int nbands_ = (g.H + MORPH_RB - 1) / MORPH_RB;
                                        const unsigned per_frame_ = (unsigned)(nbands_ * g.WW);
                                        unsigned* bitmap = blockcnt + (size_t)g.Kcap * per_frame_ + (size_t)i * block_bitmap_stride(per_frame_);
                                        const int y = gwr / g.WW, w = gwr - y * g.WW;
                                        const int by = y / MORPH_RB, ry = y - by * MORPH_RB;
                                        const int dy0 = (ry < 2 && by > 0) ? -1 : 0, dy1 = (ry >= MORPH_RB - 2 && by + 1 < nbands_) ? 1 : 0;
                                        const int dx0 = (q == 0 && (bits & 0x7u) && w > 0) ? -1 : 0;
                                        const int dx1 = (q == LPW - 1 && (bits & (7u << (PPL - 3))) && w + 1 < g.WW) ? 1 : 0;
                                        for (int dy = dy0; dy <= dy1; ++dy)
                                            for (int dx = dx0; dx <= dx1; ++dx) {
                                                const unsigned blk = (unsigned)((by + dy) * g.WW + w + dx);
                                                atomicOr(&bitmap[blk >> 5], 1u << (blk & 31));
                                            }
                                    }
                                }
                            }
#endif
                        } else {
                        if (q == 0 && gw < words && v) {  // rare: only words with set bits are stored (tagged with their frame)
                            // (the word index through an opaque move: everything derived from it -- the store's address, the block
                            // number, the bitmap's dword and bit -- is otherwise hoisted out of the frame loop as per-lane constants
                            // and held in registers for the whole launch: 154 instead of 148 at n = 8)
                            int gwr = gw;
                            asm volatile("" : "+v"(gwr));
                            bbits[(size_t)i * words + gwr] = ActWord{v, (u64)(g.t_base + 1 + i)};
                            // (LEAN: one frame, its morphology looks at the frame's block counts directly: no list)
                            if (blockcnt != nullptr) note_block<!LEAN>(blockcnt, g, i, gwr, (unsigned)__popcll(v));  // rare
                        }
                        }
                    }
                }
            }
            if constexpr (!PINGPONG) {
#pragma unroll
            for (int u = 0; u < U; ++u) cur[u] = nxt[u];
            if constexpr (TAB) {
#pragma unroll
                for (int u = 0; u < U; ++u) wcur[u] = wnxt[u];
            }
            }
        });
        if constexpr (TAB) {
            pt += N;
            wr += N;
        }
    }
#undef BSTART
#undef BCNT
    if constexpr (LEAN) lean_wave_done(lean, gridDim.x * gridDim.y * (blockDim.x >> 6));
    if constexpr (DENSE) {
        if (nzwords != nullptr && lane == 0 && nz_mine) atomicAdd(nzwords, (unsigned long long)nz_mine);
    }
#ifndef WIDE_PIN_V
#define WIDE_PIN_V "v135"  // (A/B builds: -DWIDE_PIN_V='"v151"' -- round 5's 152 registers)
#endif
#if WIDE_PIN152
    // With the pass tables the n = 7 .. 9 instantiations need 118-140 registers: FOUR waves per SIMD, which leave a Hough wave (54
    // registers) no room beside them -- 960x540 at the pipeline's pace 8.5 M -> 7.5 M frames/s (gpurun_out/r06_04).  Naming a high
    // register as clobbered holds the kernel's allocation up.  First at 152 (round 5's figure: three window waves and 56 registers for
    // ONE Hough wave); but the stages queued beside a window launch are not all that small -- k_noise_finalize, which every window
    // launch waits for, takes 64 registers, k_morph_close_dense 85, k_morph_close_t 52 -- and a kernel that finds no room on any SIMD
    // waits for a window workgroup to leave.  At 136 (the smallest allocation that still means three waves) 104 registers per SIMD
    // stay free: 1080p / n = 8 at the pipeline's pace +4 % on the same box, 960x540 +3-5 % (gpurun_out/r06_41).
    // (n = 17, 18 with 8 pixels per lane: 122-128 registers, the same four waves)
    if constexpr (TAB && ((!DENSE && PPL == 16 && N >= 7 && N <= 9) || (PPL == 8 && (N == 17 || N == 18)))) asm volatile("" ::: WIDE_PIN_V);
#endif
}
// waves per SIMD the instantiation is held to (0: whatever its natural register use allows).  Only the raw-window build asks: its
// n = 16 kernel lands three registers above the 168 that three waves per SIMD allow (A/B builds: -DWIDE_RAW_MINN=2 -DWIDE_RAW_OCC=1)
#ifndef WIDE_RAW_OCC
#define WIDE_RAW_OCC 0
#endif
template <int N, int PPL, bool LEAN>
constexpr int wide_min_waves() {
    return (WIDE_RAW_OCC != 0 && WIDE_MAX3 != 0 && !LEAN && PPL == 16 && N >= WIDE_RAW_MINN) ? (N <= 8 ? 4 : (N <= 16 ? 3 : 2)) : 0;
}
template <int N, int PPL, bool LEAN, bool DENSE = false>
__global__ void
#if WIDE_RAW_OCC
__launch_bounds__(256, (wide_min_waves<N, PPL, LEAN>()))
#else
__launch_bounds__(256)
#endif
WIDE_OCC k_stack_threshold_wide(const uint8_t* const* __restrict__ ptrtab,
                                                              const int* __restrict__ thr_tab, ActWord* __restrict__ bbits,
                                                              unsigned* __restrict__ blockcnt, PassGeom g, int Tc,
                                                              LeanArgs lean, unsigned long long* __restrict__ nzwords,
                                                              const WinRec* __restrict__ wrec) {
    stack_threshold_wide_body<N, PPL, LEAN, DENSE>(ptrtab, thr_tab, bbits, blockcnt, g, Tc, lean, nzwords, wrec);
}

// The per-frame protocol's window kernel, launched AHEAD of its frame (round 3).  A frame's chain used to start with the host's
// launch call and the command processor's dispatch -- 3 us of host time plus 3-5 us until the first wave runs, with the GPU
// idle: in the reference's loop nothing else is in flight.  mdh_update now queues the NEXT frame's window kernel behind the
// current frame's last launch; it becomes resident when that launch ends (while the host still unpacks the results) and waits
// for a MAILBOX in pinned host memory.  The next mdh_update does its bookkeeping as before and then, instead of launching,
// writes the kernel's arguments and a sequence number into the mailbox: the first loads of the frame are a PCIe poll period
// away.  One lane of workgroup (0, 0) polls the host word (2000 waves polling across the bus would saturate it) and republishes
// it in a device word the other workgroups poll.  The kernel gives up -- cancelled by the host (anything that is not the
// expected direct frame: a batched pass, a frame in pageable memory, get_binary, destroy ...) or after MAIL_EXPIRE_US without a
// frame -- and says so in the mailbox; the decision is taken by that one lane, so a frame is never processed twice.
struct MailArgs {
    const uint8_t* const* ptrtab;
    const int* thr_tab;
    ActWord* bbits;
    unsigned* blockcnt;
    PassGeom g;
    int Tc;
    LeanArgs lean;
};
// Mailbox layout: MAIL_CHUNKS lines of 64 bytes, each [15 payload dwords | tag].  The host writes the payload -- the kernel's
// arguments, then the pass's pointer table (n entries) -- and then every used line's tag = sequence number; the device reads a
// line with one 64-byte bus read, so a line whose tag is right carries its payload (x86 stores become visible in order): seeing
// the frame costs the kernel ONE bus round trip, arguments and table included.  `~s` in line 0's tag cancels kernel s.
#define MAIL_CHUNKS 8
#define MAIL_PAYLOAD (15 * MAIL_CHUNKS)
struct __attribute__((aligned(64))) MailBox {
    unsigned line[MAIL_CHUNKS][16];  // host -> kernel
    unsigned state;                  // kernel -> host: `s` = taken, `~s` = given up (expired or cancelled)
    unsigned pad1[15];
};
// tag of a mailbox line: the sequence number plus a weighted sum of the line's payload (see the poll loop of k_stack_threshold_mail)
__host__ __device__ inline unsigned mail_line_tag(const unsigned* payload15, unsigned seq) {
    unsigned c = 0;
    for (int k = 0; k < 15; ++k) c += payload15[k] * (2u * (unsigned)k + 1u);
    return seq + c;
}
#define MAIL_ARG_DW ((int)((sizeof(MailArgs) + 7) / 8) * 2)  // dwords of the arguments in the payload (the table follows, 8-byte aligned)
// What k_hough_single zeroes behind its results (clean-after-use) the waiting kernel can zero instead, in front of its poll and
// with every workgroup: the previous frame's last launch then ends with its results, and this kernel is resident that much earlier.
struct MailClean {
    int* rowcount;   // [H]
    int* litcount;   // [1]
    int* bbox;       // [4]
    long long* sums; // [2] (may be null)
    unsigned* blockcnt;
    unsigned n_blockcnt;
    int H;
};
#define MAIL_EXPIRE_TICKS 15000  // 150 us of the 100 MHz wall clock
template <int N, int PPL = 16>
__global__ void __launch_bounds__(256) WIDE_OCC k_stack_threshold_mail(MailBox* __restrict__ mail, unsigned seq, unsigned* __restrict__ go, MailClean cl) {
    if (cl.blockcnt != nullptr) {  // the previous frame's slot, left as a fresh one's (that frame's launches are done: same stream)
        const unsigned gt = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x, gn = gridDim.x * gridDim.y * blockDim.x;
        for (unsigned e = gt; e < cl.n_blockcnt; e += gn) cl.blockcnt[e] = 0;
        for (unsigned e = gt; e < (unsigned)cl.H; e += gn) cl.rowcount[e] = 0;
        if (gt == 0) cl.litcount[0] = 0;
        if (gt < 4) cl.bbox[gt] = 0;
        if (gt < 2 && cl.sums) cl.sums[gt] = 0;
    }
    // go[0]: the sequence number republished in device memory; go[32 ..]: the payload, copied across the bus ONCE by the polling
    // wave (2000 waves fetching their arguments from host memory themselves made the kernel 70 us slower than the launch it
    // replaces; an agent-scope acquire per wave behind the poll, 13 us: 2000 L2 invalidations)
    __shared__ unsigned s_go;
    constexpr int NCH = (MAIL_ARG_DW + 2 * N + 14) / 15;  // lines in use
    static_assert(NCH <= MAIL_CHUNKS, "mailbox too small");
    unsigned* const dpay = go + 32;
    if (threadIdx.x < 64) {
        unsigned v = 0;
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            const int lane = threadIdx.x;
            const unsigned* src = &mail->line[0][0];
            unsigned w0 = 0, w1 = 0;  // dwords lane and 64 + lane of the mailbox
            const long long t0 = wall_clock64();
            // two reads of the mailbox in flight, half a bus round trip apart: a frame is seen on average a quarter of a round
            // trip after it was written instead of a half
            unsigned n0 = __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            unsigned n1 = __hip_atomic_load(src + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __builtin_amdgcn_s_sleep(12);
            for (;;) {
                const unsigned m0 = __hip_atomic_load(src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned m1 = __hip_atomic_load(src + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w0 = n0; w1 = n1;  // (the older pair: the compiler waits for it with the younger pair still in flight)
                n0 = m0; n1 = m1;
                const bool tag_lane = (lane & 15) == 15;
                // A line's tag is seq + checksum(its 15 payload dwords) (weights 1, 3, 5 ...: mail_line_tag on the host), so a line
                // whose tag arrived ahead of its payload -- nothing in the ISA or the bus promises that a wave-wide 4-byte-per-lane
                // load returns a 64-byte line as ONE read -- does not pass for a frame: it is simply read again.
                unsigned c0 = tag_lane ? 0u : w0 * (2u * (unsigned)(lane & 15) + 1u), c1 = tag_lane ? 0u : w1 * (2u * (unsigned)(lane & 15) + 1u);
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    c0 += (unsigned)__shfl_xor((int)c0, o);
                    c1 += (unsigned)__shfl_xor((int)c1, o);
                }
                const u64 ok0 = __ballot(tag_lane && w0 - c0 == seq), ok1 = __ballot(tag_lane && w1 - c1 == seq);
                const unsigned tag0 = __builtin_amdgcn_readlane(w0, 15), sum0 = __builtin_amdgcn_readlane(c0, 15);
                // tags of lines 0 .. NCH-1: lanes 15, 31, 47, 63 of w0, then of w1
                int good = 0;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) good += (int)(((ch < 4 ? ok0 : ok1) >> (16 * (ch & 3) + 15)) & 1);
                if (good == NCH) { v = seq; break; }
                // (cancelled -- or the mailbox already speaks to this kernel's successor: cancelled before it became resident)
                if (tag0 == ~seq || tag0 - sum0 == seq + 1 || tag0 == ~(seq + 1)) { v = ~seq; break; }
                if (wall_clock64() - t0 > MAIL_EXPIRE_TICKS) { v = ~seq; break; }
            }
            if (v == seq) {  // payload dword p of line ch sits at mailbox dword 16 ch + p -> device dword 15 ch + p
                if ((lane & 15) != 15) __hip_atomic_store(dpay + 15 * (lane >> 4) + (lane & 15), w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((lane & 15) != 15) __hip_atomic_store(dpay + 15 * (4 + (lane >> 4)) + (lane & 15), w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) {
                __hip_atomic_store(go, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&mail->state, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        } else if (threadIdx.x == 0) {
            for (;;) {
                v = __hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v == seq || v == ~seq) break;
                __builtin_amdgcn_s_sleep(4);
            }
        }
        if (threadIdx.x == 0) s_go = v;
    }
    __syncthreads();
    if (s_go != seq) return;
    // (no cache maintenance: nothing of this kernel has read the payload's lines before the poll ended, the poller wrote them
    // through, and the lines a previous frame's kernel may have left in this XCD's L2 went at this kernel's dispatch)
    asm volatile("" ::: "memory");
    const MailArgs* a = (const MailArgs*)dpay;
    const PassGeom g = a->g;
    const LeanArgs lean = a->lean;
    // (a->ptrtab is the host's to set: the table's copy behind the arguments, dpay + MAIL_ARG_DW)
    stack_threshold_wide_body<N, PPL, true, false>(a->ptrtab, a->thr_tab, a->bbits, a->blockcnt, g, a->Tc, lean, nullptr);
}

#ifndef MDH_WINDOW_ONLY  // (to the end of the file: everything that is not a window kernel)
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

// load_win on a row of a tagged frame; `row` points at word 0 of image row y
__device__ __forceinline__ u128 load_win_act(const ActWord* __restrict__ row, int w, int WW, u64 t) {
    const u64 c = act_word(row, w, t);
    const u64 l = w > 0 ? act_word(row, w - 1, t) : 0ull;
    const u64 r = w + 1 < WW ? act_word(row, w + 1, t) : 0ull;
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

// CANDIDATE BLOCKS of a dense pass (round 5).  The median keeps a pixel only with >= 5 set samples in its 3x3 window, i.e. only
// where SOME ROW holds two set bits at most two columns apart (at the left / right image border a single bit counts twice: the
// border is replicated).  On the noise a lowered threshold lets through (8K at m3det_high: 9 % of the words carry a bit, 0.03 %
// such a pair) that names a few hundred of a frame's 65 000 blocks: this kernel streams once through the plain words the window
// kernel wrote (8 bytes per lane, a wave per image row, four rows in flight) and flags those blocks in the sparse format's block
// bitmap; k_morph_select_touched<false> lists their 3x3 block neighbourhoods and k_morph_close_t<true> closes them from the plain
// words -- instead of k_morph_close_plain's 22 dependent row loads per 16 output rows over EVERY word (0.96 ms of a 4.5 ms pass).
// A pair across two words: the left word's top two bits come over a DPP wave shift (lane 0 fetches its neighbour's word).
// (Tried first inside the window kernel, on the word it has just gathered: ~15 vector instructions per wave-frame on a kernel
// whose vector-memory front end is the bound -- 3.50 -> 3.96 ms per 500 8K frames, more than the morphology had to give.)
__global__ void __launch_bounds__(256) k_dense_candidates(const u64* __restrict__ bplain, unsigned* __restrict__ blockcnt, PassGeom g) {
    RAISE_WAVE_PRIORITY();
    const int lane = threadIdx.x & 63;
    const int nbands = (g.H + MORPH_RB - 1) / MORPH_RB;
    const unsigned per_frame = (unsigned)(nbands * g.WW);
    unsigned* bitmap = blockcnt + (size_t)g.Kcap * per_frame;
    const unsigned stride = block_bitmap_stride(per_frame);
    const unsigned rows_total = (unsigned)g.K * (unsigned)g.H;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    const int pl = (g.W - 1) & 63;  // position of pixel W - 1 in the last word
    constexpr int RB = 4;           // rows a wave has in flight
    for (unsigned r0 = wave * RB; r0 < rows_total; r0 += nwaves * RB) {
        for (int w0 = 0; w0 < g.WW; w0 += 64) {
            const int w = w0 + lane;
            u64 v[RB];
            u64 lw[RB];  // lane 0: the word left of the wave's first
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const unsigned rt = r0 + (unsigned)k;
                const u64* row = bplain + (size_t)rt * g.WW;
                v[k] = (rt < rows_total && w < g.WW) ? row[w] : 0ull;
                lw[k] = (rt < rows_total && lane == 0 && w0 > 0) ? row[w0 - 1] : 0ull;
            }
#pragma unroll
            for (int k = 0; k < RB; ++k) {
                const unsigned rt = r0 + (unsigned)k;
                const unsigned top2 = (unsigned)(v[k] >> 62);
                unsigned lt = (unsigned)__builtin_amdgcn_update_dpp(0, (int)top2, 0x138, 0xF, 0xF, false);  // wave_shr:1  lane l <- l - 1
                if (lane == 0) lt = (unsigned)(lw[k] >> 62);
                const unsigned lo2 = (unsigned)v[k] & 3u;
                bool cand = (v[k] & ((v[k] >> 1) | (v[k] >> 2))) != 0;
                cand = cand || ((lo2 & 1u) && lt) || ((lo2 & 2u) && (lt & 2u));
                cand = cand || (w == 0 && (v[k] & 1ull)) || (w == g.WW - 1 && ((v[k] >> pl) & 1ull));
                if (__ballot(cand) != 0) {
                    if (cand) {
                        const unsigned i = rt / (unsigned)g.H, y = rt - i * (unsigned)g.H;
                        const unsigned blk = (y / MORPH_RB) * (unsigned)g.WW + (unsigned)w;
                        atomicOr(&bitmap[(size_t)i * stride + (blk >> 5)], 1u << (blk & 31));
                    }
                }
            }
        }
    }
}

// a5 + a6 + a7: act = MORPH_CLOSE( median3(diff) > T ) computed from b = (diff > T):
//   median3(diff) > T  <=>  at least 5 of the 9 (border-replicated) neighbours have diff > T.
// Work selection: a pixel survives the median only with >= 5 set samples in its 3x3 window, and every stencil reaches at
// most 3 rows / 3 columns, so if the 3x3 neighbourhood of (8-row band, word column) blocks of b holds fewer than 5 set
// bits (border replication can count a sample twice per axis) the whole output block is zero and nothing is stored
// for it (absent words read as zero).  The blocks that may produce output are compacted into a work list, so that
// the stencil kernel runs with full waves instead of a few live lanes per wave.
// (device function: frame i, block `tid` of the frame; whole waves call it together -- k_post_single walks a frame with one workgroup)
template <bool COUNTED = true>
__device__ __forceinline__ bool morph_block_keep(const unsigned* __restrict__ blockcnt, const PassGeom& g, int i, int tid) {
    const int nb = (g.H + MORPH_RB - 1) / MORPH_RB;
    bool keep = false;
    // (COUNTED = false: the bitmap names candidate blocks of a DENSE pass -- there are no counts: every block of a candidate's
    // 3x3 neighbourhood is listed)
    if (!COUNTED) return tid >= 0 && tid < nb * g.WW;
    if (tid < nb * g.WW) {
        const int band = tid / g.WW, w = tid - band * g.WW;
        const int y0 = band * MORPH_RB;
        unsigned total = 0;
        for (int db = -1; db <= 1; ++db) {
            const int bb = band + db;
            if (bb < 0 || bb >= nb) continue;
            const unsigned* f = blockcnt + ((size_t)i * nb + bb) * g.WW;
            total += f[w];
            if (w > 0) total += f[w - 1];
            if (w + 1 < g.WW) total += f[w + 1];
        }
        const unsigned my = (y0 - 3 < 0 || y0 + MORPH_RB + 2 > g.H - 1) ? 2u : 1u;
        const unsigned mx = (w == 0 || 64 * (w + 1) + 3 > g.W - 1) ? 2u : 1u;
        keep = total * my * mx >= 5u;
    }
    return keep;
}
// The work list is kept PER FRAME (segment i = entries [i * blocks_per_frame, ...), count in workcount[i]).  One list with one
// counter serialised the selection on a dense binary (a lowered threshold lets noise through: every block of every frame is kept):
// 130 k same-address atomics, 1.35 ms per 128 frames at 8K.  With a counter per frame the appends of different frames go to
// different L2 channels.
// Round 4: the candidates are the 3x3 neighbourhoods of the TOUCHED blocks (note_block's bitmap), not every block of the frame.
// One wave per frame; a lane takes a dword of the bitmap (32 blocks of one band), dilates it by one block in both directions from
// the dwords around it, and the wave evaluates the candidate blocks one per lane and round exactly as morph_block_keep does.
// wpf (1, 2 or 4): waves per frame -- the 64-dword rounds of a frame's bitmap are dealt out to them (960x540: 32 dwords, one round,
// one wave; 4K: 507 dwords, eight rounds, and a long streak's candidates on top: one wave per frame made the launch as long as
// its longest frame, 0.19 ms per 1000-frame pass).  The waves of a frame share nothing but the frame's list counter.
template <bool COUNTED>
__global__ void __launch_bounds__(256) k_morph_select_touched(const unsigned* __restrict__ blockcnt, unsigned* __restrict__ worklist,
                                                              unsigned* __restrict__ workcount, PassGeom g, int wpf, int dilate) {
    // dilate = 0 (COUNTED = false only): the bitmap already names every block that can produce output (flagged with their
    // neighbours where needed by the window kernel's candidate test): the list is the bitmap
    RAISE_WAVE_PRIORITY();
    const int nb = (g.H + MORPH_RB - 1) / MORPH_RB;
    const unsigned per_frame = (unsigned)(nb * g.WW);
    const int wv = (int)(threadIdx.x >> 6);
    const int i = blockIdx.x * (4 / wpf) + wv / wpf;
    const unsigned sub = (unsigned)(wv % wpf);
    if (i >= g.K) return;
    const int lane = threadIdx.x & 63;
    const unsigned stride = block_bitmap_stride(per_frame);
    const unsigned* bm = blockcnt + (size_t)g.Kcap * per_frame + (size_t)i * stride;
    unsigned cnt = 0;
    for (unsigned d = (unsigned)lane; d < stride; d += 64) cnt += (unsigned)__popc(bm[d]);
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (cnt == 0) return;
    auto append = [&](bool keep, int blk) {
        const u64 m = __ballot(keep);
        if (!m) return;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(&workcount[(size_t)i * WORKCNT_STRIDE], (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (keep) worklist[(size_t)i * per_frame + base + __popcll(m & ((1ull << lane) - 1))] = (unsigned)i * per_frame + (unsigned)blk;
    };
    if (9u * cnt >= per_frame) {
        // a frame with most of its blocks touched (a binary with noise in it, before the pass format flips to dense): every block
        // tests itself, as round 3's selection did for all frames
        for (unsigned b0 = 64 * sub; b0 < per_frame; b0 += 64 * (unsigned)wpf) {
            const int tid = (int)b0 + lane;
            append(morph_block_keep<COUNTED>(blockcnt, g, i, tid), tid);
        }
        return;
    }
    // 32 bitmap bits starting at bit position pos of the frame's bitmap (positions outside [0, 32 * stride) read as zero)
    auto bits_at = [&](int pos) -> unsigned {
        const int d = pos >> 5, sh = pos & 31;  // (arithmetic shift: d = floor(pos / 32))
        const unsigned lo = (d >= 0 && d < (int)stride) ? bm[d] : 0u;
        const unsigned hi = (d + 1 >= 0 && d + 1 < (int)stride) ? bm[d + 1] : 0u;
        return sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
    };
    // The candidates of 64 bitmap dwords are first compacted into the wave's queue in LDS (entry = lane << 5 | bit), then evaluated
    // 64 at a time: a streak's candidates sit in a few dwords (a dozen bits each in two or three lanes), and evaluated straight from
    // the lanes' own masks the wave ran a dozen rounds of nine dependent loads for three live lanes each.
    // A block is a candidate when a touched block sits at distance -1, 0, +1 in the linear numbering of its own band or of the bands
    // above and below: at the ends of a band that also names the neighbouring band's far end -- a few candidates too many, which
    // morph_block_keep (the exact test, on the counts) turns down like any other.
    __shared__ unsigned short s_q[4][2048];
    unsigned short* q = s_q[threadIdx.x >> 6];
    for (unsigned d0 = 64 * sub; d0 < stride; d0 += 64 * (unsigned)wpf) {
        const unsigned d = d0 + (unsigned)lane;
        unsigned cand = 0;
        if (d < stride) {
            const int p = 32 * (int)d;
            if (dilate) {
                for (int dy = -g.WW; dy <= g.WW; dy += g.WW) cand |= bits_at(p + dy - 1) | bits_at(p + dy) | bits_at(p + dy + 1);
            } else {
                cand = bm[d];
            }
            const int left = (int)per_frame - p;  // blocks the dword covers
            if (left < 32) cand &= (1u << left) - 1u;
        }
        if (__ballot(cand != 0) == 0) continue;
        const unsigned pc = (unsigned)__popc(cand);
        unsigned incl = pc;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(incl, o);
            if (lane >= o) incl += t;
        }
        const unsigned total = __shfl(incl, 63);
        unsigned pos = incl - pc;
        while (cand) {
            q[pos++] = (unsigned short)((lane << 5) | __builtin_ctz(cand));
            cand &= cand - 1u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (one wave: its LDS operations execute in order)
        __builtin_amdgcn_wave_barrier();
        for (unsigned e0 = 0; e0 < total; e0 += 64) {
            const unsigned e = e0 + (unsigned)lane;
            const bool have = e < total;
            const unsigned v = have ? q[e] : 0u;
            const int tid = 32 * (int)(d0 + (v >> 5)) + (int)(v & 31u);
            append(have && morph_block_keep<COUNTED>(blockcnt, g, i, tid), tid);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// One thread owns one block (a word column x a band of ROWS rows) and slides down it with 3-row pipelines.  `load_row(yc)` returns
// the 128-bit window of image row yc around word column w (load_win_act's layout); `live` = the thread's column exists (the dense
// kernel keeps the lanes beyond the last column in step for its lane exchanges).
template <int ROWS, class Loader>
__device__ __forceinline__ void morph_close_band(Loader&& load_row, int i, int w, int y0, bool live, ActWord* __restrict__ act_ring,
                                                 unsigned* __restrict__ nzlist, unsigned* __restrict__ nzcount, const PassGeom& g) {
    const Win q = make_win(w, g.W);
    const long long t = g.t_base + 1 + i;
    ActWord* dst = act_ring + (size_t)((t - 1) % g.R) * g.H * g.WW;
    // The median of a row needs >= 5 set samples in some 3x3 window, i.e. >= 5 set bits in the three 128-bit row windows it is
    // taken from (>= 3 next to the left / right image border, where replication counts a pixel twice; rows replicated at the top
    // / bottom border are loaded twice and counted twice anyway).  On noise -- a lowered threshold lets half a per cent of the
    // pixels through, and then every block of a frame is listed -- 24 of 25 rows fail that test, and the bit-sliced 3x3 count
    // (two thirds of the work of an item) is only run for the rest.  The raw rows are kept instead of their horizontal sums.
    u128 bA = 0, bB = 0;
    int pA = 0, pB = 0;
    u128 mA = 0, mB = 0;
    u128 dA = 0, dB = 0;
    const u128 ones = ~(u128)0;
    const int mx = (w == 0 || 64 * (w + 1) + 3 > g.W - 1) ? 2 : 1;
#pragma unroll
    for (int it = 0; it < ROWS + 6; ++it) {
        const int yy = y0 - 3 + it;
        const int yc = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy);
        u128 mC = 0, dC = 0;
        const u128 brow = load_row(yc);
        const int pC = __popcll((u64)brow) + __popcll((u64)(brow >> 64));
        const int ym = yy - 1;  // row whose median is complete now
        const int yd = yy - 2;
        if ((pA + pB + pC) * mx >= 5 && ym >= 0 && ym < g.H) {  // (a row outside the image never wins the dilate: its median stays 0)
            u128 hA0, hA1, hB0, hB1, hC0, hC1;
            hsum3_replicate(bA, q, hA0, hA1);
            hsum3_replicate(bB, q, hB0, hB1);
            hsum3_replicate(brow, q, hC0, hC1);
            mC = count_ge5(hA0, hA1, hB0, hB1, hC0, hC1) & q.valid;
        }
        // sparse images: when no median row of the neighbourhood has a set pixel the dilate produces zeros
        if ((mA | mB | mC) != 0) dC = (hor_or3(mA) | hor_or3(mB) | hor_or3(mC)) & q.valid;
        const bool dC_in = yd >= 0 && yd < g.H;
        u128 eC = dC_in ? hor_and3_open(dC, q) : ones;  // rows outside the image never win the erode
        const int ya = yy - 3;
        {
            u64 a = 0;
            if (live && ya >= y0 && ya < g.H) a = (u64)((dA & dB & eC & q.valid) >> 3);
            // Only words with lit pixels are stored, and listed for the sparse dynamic-mask pass: one append per wave and row step
            // (the lanes of a wave walk their items in lockstep), not one same-address atomic per word.
            const u64 have = __ballot(a != 0);
            if (have) {
                const int lane_ = threadIdx.x & 63;
                const int leader = __builtin_ctzll(have);
                unsigned base = 0;
                if (lane_ == leader) base = atomicAdd(nzcount, (unsigned)__popcll(have));
                base = __shfl(base, leader);
                if (a) {
                    dst[(size_t)ya * g.WW + w] = ActWord{a, (u64)t};
                    nzlist[base + __popcll(have & ((1ull << lane_) - 1))] = (unsigned)((size_t)i * g.H * g.WW + (size_t)ya * g.WW + w);
                }
            }
        }
        bA = bB; bB = brow;
        pA = pB; pB = pC;
        mA = mB; mB = mC;
        dA = dB; dB = eC;
    }
}
__device__ __forceinline__ void morph_close_item(const ActWord* __restrict__ bbits, unsigned item, ActWord* __restrict__ act_ring,
                                                 unsigned* __restrict__ nzlist, unsigned* __restrict__ nzcount, const PassGeom& g) {
    const int nb = (g.H + MORPH_RB - 1) / MORPH_RB;
    const int i = item / (unsigned)(nb * g.WW);
    const int tid = item - (unsigned)i * (unsigned)(nb * g.WW);
    const int band = tid / g.WW, w = tid - band * g.WW;
    const ActWord* src = bbits + (size_t)i * g.H * g.WW;
    const u64 t = (u64)(g.t_base + 1 + i);
    morph_close_band<MORPH_RB>([&](int yc) { return load_win_act(src + (size_t)yc * g.WW, w, g.WW, t); }, i, w, band * MORPH_RB, true,
                               act_ring, nzlist, nzcount, g);
}
// Dense binaries (8K at m3det_high: noise gets through the lowered threshold and EVERY block of a frame is listed).  The list
// kernel then fetches each word of b five times (three tagged 16-byte loads per row step, 14 row steps per 8 output rows) and is
// bound by those loads.  A frame with at least half of its blocks listed is left to this kernel instead: a wave takes 64
// ADJACENT word columns of a band of 16 rows, so a lane loads its own word only and gets the three bits it needs from either
// neighbour with one DPP wave shift each (lanes 0 and 63 fetch theirs), and the halo is 6 rows in 22 instead of 6 in 14: 1.4
// loads per word instead of 5.25.  Blocks the list would not have named produce zeros here and store nothing -- same result.
#define MORPH_DENSE_ROWS 16
__global__ void __launch_bounds__(64) k_morph_close_dense(const ActWord* __restrict__ bbits, const unsigned* __restrict__ dense,
                                                          ActWord* __restrict__ act_ring, unsigned* __restrict__ nzlist,
                                                          unsigned* __restrict__ nzcount, PassGeom g) {
    // dense[0] = number of dense frames of the pass, dense[1..] = their indices (written by k_morph_close, which looks at every
    // frame's count anyway): a pass without such frames costs this launch one load per wave
    const unsigned nd = dense != nullptr ? dense[0] : (unsigned)g.K;  // (null: every frame of the pass -- MDH_ALL_DENSE experiment)
    if (nd == 0) return;
    RAISE_WAVE_PRIORITY();
    const int lane = threadIdx.x;
    const unsigned cgroups = (unsigned)(g.WW + 63) / 64, bands = (unsigned)(g.H + MORPH_DENSE_ROWS - 1) / MORPH_DENSE_ROWS;
    const unsigned per_frame = cgroups * bands;
    for (unsigned item = blockIdx.x; item < nd * per_frame; item += gridDim.x) {
        const unsigned f = item / per_frame, rest = item - f * per_frame;
        const unsigned band = rest / cgroups, cg = rest - band * cgroups;
        const int i = dense != nullptr ? (int)dense[1 + f] : (int)f;
        const int w = (int)cg * 64 + lane;
        const bool live = w < g.WW;
        const int wc = live ? w : g.WW - 1;
        const ActWord* src = bbits + (size_t)i * g.H * g.WW;
        const u64 t = (u64)(g.t_base + 1 + i);
        auto load_row = [&](int yc) -> u128 {
            const ActWord* row = src + (size_t)yc * g.WW;
            const u64 c = live ? act_word(row, wc, t) : 0ull;
            // what the neighbours need of this word: its top three bits (for the lane to the right) and its low three (to the left)
            const int mine = (int)((unsigned)(c >> 61) | ((unsigned)(c & 7ull) << 3));
            const unsigned from_left = (unsigned)__builtin_amdgcn_update_dpp(0, mine, 0x138, 0xF, 0xF, false);   // wave_shr:1  lane l <- l - 1
            const unsigned from_right = (unsigned)__builtin_amdgcn_update_dpp(0, mine, 0x130, 0xF, 0xF, false);  // wave_shl:1  lane l <- l + 1
            unsigned l3 = from_left & 7u, r3 = (from_right >> 3) & 7u;
            if (lane == 0) l3 = (live && w > 0) ? (unsigned)(act_word(row, w - 1, t) >> 61) : 0u;
            if (lane == 63) r3 = (live && w + 1 < g.WW) ? (unsigned)(act_word(row, w + 1, t) & 7ull) : 0u;
            return (u128)l3 | ((u128)c << 3) | ((u128)r3 << 67);
        };
        morph_close_band<MORPH_DENSE_ROWS>(load_row, i, wc, (int)band * MORPH_DENSE_ROWS, live, act_ring, nzlist, nzcount, g);
    }
}
// Plain dense binaries (the window kernel's dense output: u64 [K][H][WW], every word written): the same
// act = close3x3(median3(b)) on 64-BIT words.  A wave takes 62 adjacent word columns of a band of MORPH_PLAIN_ROWS rows plus one
// halo column on either side (lane l <-> word w0 - 1 + l); a lane works on its own word only and gets the ONE bit a 3x3 stage
// needs from either neighbour with a DPP wave shift of its edge bits -- once per stage (b for the median, the OR of three
// median rows for the dilate, the dilated row for the erode), because a lane cannot compute its neighbour's intermediate bits
// as the 128-bit windows of morph_close_band do (64 + 3 + 3 bits, every stage on twice the registers).  The halo lanes compute
// their whole word like any other: their outer edge is wrong and never looked at, their inner edge depends on their own word and
// on lane 1 / 62 only.  Half the ALU work per word, 8 bytes loaded per word instead of 16, no tag compares.  Same row-skipping
// rule as morph_close_band (a median row needs >= 5 set samples in its three 66-bit row windows), same stores and list appends.
#define MORPH_PLAIN_ROWS 16
#define MORPH_PLAIN_COLS 62
__device__ __forceinline__ void edge_bits(u64 c, unsigned& lbit, unsigned& rbit) {
    const int mine = (int)((unsigned)(c >> 63) | ((unsigned)(c & 1ull) << 1));
    const unsigned from_left = (unsigned)__builtin_amdgcn_update_dpp(0, mine, 0x138, 0xF, 0xF, false);   // wave_shr:1  lane l <- l - 1
    const unsigned from_right = (unsigned)__builtin_amdgcn_update_dpp(0, mine, 0x130, 0xF, 0xF, false);  // wave_shl:1  lane l <- l + 1
    lbit = from_left & 1u;          // the left neighbour's top bit
    rbit = (from_right >> 1) & 1u;  // the right neighbour's bottom bit
}
__global__ void __launch_bounds__(64) k_morph_close_plain(const u64* __restrict__ bplain, const unsigned* __restrict__ frames,
                                                          ActWord* __restrict__ act_ring,
                                                          unsigned* __restrict__ nzlist, unsigned* __restrict__ nzcount, PassGeom g) {
    // frames: null = every frame of the pass; else frames[0] = how many, frames[1 ..] = which (the frames whose candidate blocks
    // cover at least half of the frame: k_morph_close_t<true> leaves those to this kernel; a pass without any costs a load per wave)
    const unsigned nf = frames != nullptr ? frames[0] : (unsigned)g.K;
    if (nf == 0) return;
    RAISE_WAVE_PRIORITY();
    const int lane = threadIdx.x;
    const unsigned cgroups = (unsigned)(g.WW + MORPH_PLAIN_COLS - 1) / MORPH_PLAIN_COLS;
    const unsigned bands = (unsigned)(g.H + MORPH_PLAIN_ROWS - 1) / MORPH_PLAIN_ROWS;
    const unsigned per_frame = cgroups * bands;
    const size_t words = (size_t)g.H * g.WW;
    for (unsigned item = blockIdx.x; item < nf * per_frame; item += gridDim.x) {
        const unsigned fi = item / per_frame, rest = item - fi * per_frame;
        const unsigned i = frames != nullptr ? frames[1 + fi] : fi;
        const unsigned band = rest / cgroups, cg = rest - band * cgroups;
        const int w = (int)cg * MORPH_PLAIN_COLS - 1 + lane;
        const bool live = w >= 0 && w < g.WW;
        const bool outl = live && lane >= 1 && lane <= MORPH_PLAIN_COLS;
        const int wc = live ? w : 0;
        const bool first_w = w == 0, last_w = w == g.WW - 1;
        const int pl = (g.W - 1) & 63;                                    // position of pixel W-1 in the last word
        const u64 vm = !live ? 0ull : ((last_w && pl != 63) ? ((2ull << pl) - 1) : ~0ull);  // pixels that exist
        const u64 top = 1ull << pl;
        const int mx = (first_w || last_w) ? 2 : 1;
        const int y0 = (int)band * MORPH_PLAIN_ROWS;
        const u64* src = bplain + (size_t)i * words;
        const long long t = g.t_base + 1 + (long long)i;
        ActWord* dst = act_ring + (size_t)((t - 1) % g.R) * words;
        u64 bA = 0, bB = 0;            // raw rows yy-2, yy-1
        unsigned eA = 0, eB = 0;       // their neighbours' edge bits: bit 0 = left neighbour's top bit, bit 1 = right neighbour's bottom bit
        int pA = 0, pB = 0;
        u64 mA = 0, mB = 0;            // median rows
        u64 dA = ~0ull, dB = ~0ull;    // horizontally eroded dilate rows
        auto hsum = [&](u64 b, unsigned e, u64& h0, u64& h1) {  // horizontal 3-sum with the image border replicated
            unsigned lb = e & 1u, rb = (e >> 1) & 1u;
            if (first_w) lb = (unsigned)(b & 1ull);
            u64 L = (b << 1) | (u64)lb, Rr = (b >> 1) | ((u64)rb << 63);
            if (last_w) Rr |= b & top;
            h0 = L ^ b ^ Rr;
            h1 = (L & b) | (L & Rr) | (b & Rr);
        };
#pragma unroll 2
        for (int it = 0; it < MORPH_PLAIN_ROWS + 6; ++it) {
            const int yy = y0 - 3 + it;
            const int yc = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy);
            const u64 brow = live ? src[(size_t)yc * g.WW + wc] : 0ull;  // (default cache policy: the window kernel has just written these words -- non-temporal loads here cost 6 %)
            unsigned lb, rb;
            edge_bits(brow, lb, rb);
            const unsigned eC = lb | (rb << 1);
            const int pC = __popcll(brow) + (int)lb + (int)rb;
            const int ym = yy - 1, yd = yy - 2, ya = yy - 3;
            u64 mC = 0;
            if ((pA + pB + pC) * mx >= 5 && ym >= 0 && ym < g.H) {  // (a row outside the image never wins the dilate: its median stays 0)
                u64 a0, a1, b0, b1, c0, c1;
                hsum(bA, eA, a0, a1);
                hsum(bB, eB, b0, b1);
                hsum(brow, eC, c0, c1);
                const u64 s0 = a0 ^ b0 ^ c0, k0 = (a0 & b0) | (a0 & c0) | (b0 & c0);
                const u64 s1 = a1 ^ b1 ^ c1, k1 = (a1 & b1) | (a1 & c1) | (b1 & c1);
                const u64 t1 = k0 ^ s1, c2 = k0 & s1, u2 = c2 ^ k1, u3 = c2 & k1;
                mC = (u3 | (u2 & (t1 | s0))) & vm;  // at least 5 of the 9 samples set
            }
            // dilate row yd: the OR of three median rows, then ONE horizontal OR-3 (OR commutes)
            const u64 v = mA | mB | mC;
            if (yd >= 0 && yd < g.H && __ballot(v != 0) == 0) {
                // (almost every row on a noisy sky: the median has removed what the threshold let through.)  No median pixel
                // anywhere in the wave's three rows: the dilated row yd -- a row of the image -- is zero in every lane, its horizontal
                // erosion zero wherever a pixel exists, so nothing can come out for row ya, and two of the three edge exchanges
                // and all of the dilate / erode logic are skipped.  (A dilate row OUTSIDE the image never wins the erode: it
                // counts as all ones, and row ya = H - 1 can still come out of the two rows above it: the general path.)
                bA = bB; bB = brow;
                eA = eB; eB = eC;
                pA = pB; pB = pC;
                mA = mB; mB = 0;
                dA = dB; dB = ~vm;
                continue;
            }
            edge_bits(v, lb, rb);
            const u64 dC = (v | (v << 1) | (v >> 1) | (u64)lb | ((u64)rb << 63)) & vm;
            // erode, horizontal part: positions outside the image count as 1; a dilate row outside the image never wins
            const bool dC_in = yd >= 0 && yd < g.H;
            const u64 e = (dC_in ? dC : ~0ull) | ~vm;  // (lanes without a word: all ones)
            edge_bits(e, lb, rb);
            if (first_w) lb = 1u;
            if (last_w) rb = 1u;
            const u64 hC = e & ((e << 1) | (u64)lb) & ((e >> 1) | ((u64)rb << 63));
            u64 a = 0;
            if (outl && ya >= y0 && ya < g.H) a = dA & dB & hC & vm;
            const u64 have = __ballot(a != 0);
            if (have) {
                const int leader = __builtin_ctzll(have);
                unsigned base = 0;
                if (lane == leader) base = atomicAdd(nzcount, (unsigned)__popcll(have));
                base = __shfl(base, leader);
                if (a) {
                    dst[(size_t)ya * g.WW + w] = ActWord{a, (u64)t};
                    nzlist[base + __popcll(have & ((1ull << lane) - 1))] = (unsigned)((size_t)i * words + (size_t)ya * g.WW + w);
                }
            }
            bA = bB; bB = brow;
            eA = eB; eB = eC;
            pA = pB; pB = pC;
            mA = mB; mB = mC;
            dA = dB; dB = hC;
        }
    }
}
// Every frame's segment of the work list gets the same share of the grid's waves (nwaves / K of them, at least one), which walk
// it 64 entries at a time: one look at the frame's count per wave.  (Walking all wave-slots of all segments instead -- a count
// load per slot -- cost the sparse 4K pipeline 10 %: 63 dependent loads per wave to find out that there is nothing to do.)
__device__ __forceinline__ void morph_close_items(const ActWord* __restrict__ bbits, const unsigned* __restrict__ worklist,
                                                  const unsigned* __restrict__ workcount, ActWord* __restrict__ act_ring,
                                                  unsigned* __restrict__ nzlist, unsigned* __restrict__ nzcount, const PassGeom& g,
                                                  unsigned first_wave, unsigned nwaves, unsigned* __restrict__ dense) {
    const unsigned per_frame = (unsigned)(((g.H + MORPH_RB - 1) / MORPH_RB) * g.WW);
    const unsigned K = (unsigned)g.K;
    const unsigned wpf = nwaves >= K ? nwaves / K : 1u;  // waves per frame
    const unsigned groups = nwaves / wpf;                // frames walked side by side
    const unsigned grp = __builtin_amdgcn_readfirstlane(first_wave / wpf);
    const unsigned sub = __builtin_amdgcn_readfirstlane(first_wave - grp * wpf);
    const unsigned lane = threadIdx.x & 63;
    if (grp >= groups) return;
    for (unsigned i = grp; i < K; i += groups) {
        const unsigned cnt = workcount[(size_t)i * WORKCNT_STRIDE];
        if (dense != nullptr && 2u * cnt >= per_frame) {  // k_morph_close_dense's frame: the frame's first wave puts it on that list
            if (sub == 0 && lane == 0) dense[1 + atomicAdd(&dense[0], 1u)] = i;
            continue;
        }
        const unsigned* seg = worklist + (size_t)i * per_frame;
        for (unsigned e = sub * 64 + lane; e < cnt; e += wpf * 64) morph_close_item(bbits, seg[e], act_ring, nzlist, nzcount, g);
    }
}
// One frame (the per-frame protocol): every block tests its own neighbourhood and goes ahead -- no work list, one launch less
// in the frame's chain of dependent commands.
__global__ void __launch_bounds__(64) k_morph_close_frame(const ActWord* __restrict__ bbits, const unsigned* __restrict__ blockcnt,
                                                          ActWord* __restrict__ act_ring, unsigned* __restrict__ nzlist,
                                                          unsigned* __restrict__ nzcount, PassGeom g) {
    RAISE_WAVE_PRIORITY();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (morph_block_keep(blockcnt, g, 0, tid)) morph_close_item(bbits, (unsigned)tid, act_ring, nzlist, nzcount, g);
}
// One frame, TRANSPOSED (round 3, the per-frame protocol): an item's fourteen input rows on fourteen LANES instead of fourteen steps of
// a one-lane pipeline.  A group of 16 lanes owns one listed block (band of 8 rows x word column); lane j holds the 128-bit window
// of row y0 - 3 + j and gets its vertical neighbours' intermediate rows with DPP row shifts (the 16 lanes of a group are one DPP
// row): median on lanes 1..12, dilate on 2..11, erode + output on 3..10.  The whole 3-stage stencil of a block is then ~250
// instructions deep instead of 14 x ~150: a streak frame's morphology (a few dozen blocks, one or two per wave: a serial chain of
// 13-30 us in k_morph_close_frame) drops to the launch's floor.  Same borders as morph_close_band: rows replicated for the median
// (clamped loads), a median row outside the image is zero, a dilate row outside the image never wins the erode.
__device__ __forceinline__ u128 dpp_row_u128(u128 v, bool from_below) {
    // lane j <- lane j - 1 (row_shr:1) of the same 16-lane row, or lane j + 1 (row_shl:1); lanes without a source get zero
    unsigned w[4] = {(unsigned)v, (unsigned)(v >> 32), (unsigned)(v >> 64), (unsigned)(v >> 96)};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        w[k] = from_below ? (unsigned)__builtin_amdgcn_update_dpp(0, (int)w[k], 0x101, 0xF, 0xF, false)   // row_shl:1
                          : (unsigned)__builtin_amdgcn_update_dpp(0, (int)w[k], 0x111, 0xF, 0xF, false);  // row_shr:1
    return (u128)w[0] | ((u128)w[1] << 32) | ((u128)w[2] << 64) | ((u128)w[3] << 96);
}
// One listed block per 16-lane group (see above).  src / dst: word 0 of the frame's b image / of its row in the ring of binaries;
// idx_base: the frame's first word in the pass-wide numbering of the non-zero list.  Whole waves call it together (DPP, ballot).
// The words that come out are stored at once; their list entries are QUEUED in the wave's LDS (q, qn) and appended by
// morph_nz_flush with ONE atomic on the pass's list counter per flush: every append of every wave of a pass meets in that one
// counter, and an atomic per group of four blocks (25 k per 16 000-frame pass at 960x540) made the launch wait on it.
#define MORPH_NZQ 512
__device__ __forceinline__ void morph_nz_flush(unsigned* __restrict__ q, unsigned& qn, unsigned* __restrict__ nzlist,
                                               unsigned* __restrict__ nzcount, int lane) {
    if (qn == 0) return;  // (wave-uniform)
    __syncthreads();      // (one wave per workgroup: orders the queue's LDS writes before the reads below)
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(nzcount, qn);
    base = __shfl(base, 0);
    for (unsigned e = (unsigned)lane; e < qn; e += 64) nzlist[base + e] = q[e];
    __syncthreads();
    qn = 0;
}
template <bool PLAIN = false>
__device__ __forceinline__ void morph_close_group_t(const ActWord* __restrict__ src, u64 t, ActWord* __restrict__ dst, unsigned idx_base,
                                                    bool have, int blk, unsigned* __restrict__ q, unsigned& qn, const PassGeom& g,
                                                    int lane, int j) {
    const u128 ones = ~(u128)0;
    const int band = blk / g.WW, w = blk - band * g.WW;
    const int y0 = band * MORPH_RB;
    const int yy = y0 - 3 + j;  // this lane's row (j = 14, 15: spare lanes, harmless)
    const int yc = yy < 0 ? 0 : (yy >= g.H ? g.H - 1 : yy);
    const Win wq = make_win(w, g.W);
    u128 b = 0;
    // (PLAIN: src is the window kernel's dense output, u64 [H][WW] of this frame, every word written: no tags)
    if (have && j < 14) b = PLAIN ? load_win((const u64*)src + (size_t)yc * g.WW, w, g.WW) : load_win_act(src + (size_t)yc * g.WW, w, g.WW, t);
    // median of this lane's row from the rows above and below (lanes j - 1, j + 1)
    u128 h0, h1;
    hsum3_replicate(b, wq, h0, h1);
    const u128 a0 = dpp_row_u128(h0, false), a1 = dpp_row_u128(h1, false);
    const u128 c0 = dpp_row_u128(h0, true), c1 = dpp_row_u128(h1, true);
    u128 med = 0;
    if (yy >= 0 && yy < g.H && j >= 1 && j <= 12) med = count_ge5(a0, a1, h0, h1, c0, c1) & wq.valid;
    // dilate
    const u128 dil = hor_or3(dpp_row_u128(med, false) | med | dpp_row_u128(med, true)) & wq.valid;
    // erode: horizontal part per row (a dilate row outside the image never wins), then the three rows
    const u128 e = (yy >= 0 && yy < g.H) ? hor_and3_open(dil, wq) : ones;
    const u128 ea = dpp_row_u128(e, false), ec = dpp_row_u128(e, true);
    u64 a = 0;
    if (have && j >= 3 && j <= 10 && yy < g.H) a = (u64)((ea & e & ec & wq.valid) >> 3);
    const u64 havem = __ballot(a != 0);
    if (havem) {
        if (a) {
            dst[(size_t)yy * g.WW + w] = ActWord{a, t};
            q[qn + __popcll(havem & ((1ull << lane) - 1))] = idx_base + (unsigned)((size_t)yy * g.WW + w);
        }
        qn += (unsigned)__popcll(havem);
    }
}
__global__ void __launch_bounds__(64) k_morph_close_frame_t(const ActWord* __restrict__ bbits, const unsigned* __restrict__ blockcnt,
                                                            ActWord* __restrict__ act_ring, unsigned* __restrict__ nzlist,
                                                            unsigned* __restrict__ nzcount, PassGeom g) {
    RAISE_WAVE_PRIORITY();
    __shared__ unsigned q[MORPH_NZQ];
    unsigned qn = 0;
    const int lane = threadIdx.x, grp = lane >> 4, j = lane & 15;
    const int nb = (g.H + MORPH_RB - 1) / MORPH_RB;
    const long long t = g.t_base + 1;
    ActWord* dst = act_ring + (size_t)((t - 1) % g.R) * g.H * g.WW;
    for (int b0 = blockIdx.x * 64; b0 < nb * g.WW; b0 += gridDim.x * 64) {
        u64 todo = __ballot(morph_block_keep(blockcnt, g, 0, b0 + lane));
        while (todo) {  // four listed blocks at a time, one per 16-lane group
            u64 m = todo;
            for (int k = 0; k < grp; ++k) m &= m - 1;
            const bool have = m != 0;
            const int blk = b0 + (have ? __builtin_ctzll(m) : 0);
            for (int k = 0; k < 4 && todo; ++k) todo &= todo - 1;
            morph_close_group_t(bbits, (u64)t, dst, 0u, have, blk, q, qn, g, lane, j);
            if (qn > MORPH_NZQ - 64) morph_nz_flush(q, qn, nzlist, nzcount, lane);
        }
    }
    morph_nz_flush(q, qn, nzlist, nzcount, lane);
}
// The batched passes' list kernel, TRANSPOSED as well (round 4).  k_morph_close gives a listed block to ONE lane, which walks its
// fourteen input rows as fourteen dependent steps of three tagged loads each; on a quiet sky a pass lists a few dozen blocks for
// one frame in three and none for the others, so the launch is a handful of lanes per wave on a serial chain of ~30 us: 0.58 ms
// per 16 000-frame pass at 960x540, a third of the window kernel's time, at a few per cent of the machine.  Here a wave takes a
// frame and four of its listed blocks at a time, a block's fourteen rows on fourteen lanes of a 16-lane group: sixteen times the
// waves, each a fraction as long.  Frames with at least half of their blocks listed go to k_morph_close_dense as before.
template <bool PLAIN>
__global__ void __launch_bounds__(64) k_morph_close_t(const ActWord* __restrict__ bbits, const unsigned* __restrict__ worklist,
                                                      const unsigned* __restrict__ workcount, ActWord* __restrict__ act_ring,
                                                      unsigned* __restrict__ nzlist, unsigned* __restrict__ nzcount, PassGeom g,
                                                      unsigned* __restrict__ dense, int wpf) {
    RAISE_WAVE_PRIORITY();
    __shared__ unsigned q[MORPH_NZQ];
    unsigned qn = 0;
    const int lane = threadIdx.x, grp = lane >> 4, j = lane & 15;
    const unsigned per_frame = (unsigned)(((g.H + MORPH_RB - 1) / MORPH_RB) * g.WW);
    const size_t words = (size_t)g.H * g.WW;
    // wpf workgroups (of one wave) per frame share the frame's list, four entries per workgroup and step
    for (unsigned u = blockIdx.x; u < (unsigned)g.K * (unsigned)wpf; u += gridDim.x) {
        const unsigned i = u / (unsigned)wpf, sub = u - i * (unsigned)wpf;
        const unsigned cnt = workcount[(size_t)i * WORKCNT_STRIDE];
        if (cnt == 0) continue;
        if (dense != nullptr && 2u * cnt >= per_frame) {
            if (lane == 0 && sub == 0) dense[1 + atomicAdd(&dense[0], 1u)] = i;
            continue;
        }
        const unsigned* seg = worklist + (size_t)i * per_frame;
        const long long t = g.t_base + 1 + (long long)i;
        const ActWord* src = PLAIN ? (const ActWord*)((const u64*)bbits + (size_t)i * words) : bbits + (size_t)i * words;
        ActWord* dst = act_ring + (size_t)((t - 1) % g.R) * words;
        for (unsigned e0 = 4 * sub; e0 < cnt; e0 += 4 * (unsigned)wpf) {
            const unsigned e = e0 + (unsigned)grp;
            const bool have = e < cnt;
            const int blk = have ? (int)(seg[e] - i * per_frame) : 0;
            morph_close_group_t<PLAIN>(src, (u64)t, dst, (unsigned)(i * words), have, blk, q, qn, g, lane, j);
            if (qn > MORPH_NZQ - 64) morph_nz_flush(q, qn, nzlist, nzcount, lane);
        }
    }
    morph_nz_flush(q, qn, nzlist, nzcount, lane);  // (one append for all the frames this wave walked)
}
__global__ void __launch_bounds__(64) k_morph_close(const ActWord* __restrict__ bbits, const unsigned* __restrict__ worklist,
                                                    const unsigned* __restrict__ workcount, ActWord* __restrict__ act_ring,
                                                    unsigned* __restrict__ nzlist, unsigned* __restrict__ nzcount, PassGeom g,
                                                    unsigned* __restrict__ dense) {
    RAISE_WAVE_PRIORITY();
    morph_close_items(bbits, worklist, workcount, act_ring, nzlist, nzcount, g, (blockIdx.x * blockDim.x + threadIdx.x) >> 6,
                      (gridDim.x * blockDim.x) >> 6, dense);
}

// a8 + a9 count.  Dynamic mask (Detector.py:234-242): a pixel is "alive" unless it was lit in ALL of the last
// len = min(n, t) binaries (sum of the last len binaries <= (len-1)*255);  dst = act & erode3x3(alive).
// Only words with lit pixels need the mask, and those are rare, so alive is evaluated on the fly from the ring of
// binaries for the 3x3 word neighbourhood instead of being materialised for the whole frame.
// Also: per-row and per-frame popcounts.
__device__ __forceinline__ u64 shfl64(u64 v, int src) {
    const unsigned lo = __shfl((unsigned)v, src), hi = __shfl((unsigned)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}

// SIXTEEN LANES PER LISTED WORD (four words per wave): lane 3 * (dy + 1) + (k + 1) of a group fetches word w + k of row
// y + dy from each of the len history frames (independent loads) and ANDs them; the 3x3 erode is finished by lanes 1,
// 4, 7 of the group (one per row) and lane 0 stores the result.
// (device function: `first_e` = this 16-lane group's first list entry, `ngroups` = groups walking the list)
// BATCH: history words fetched per round trip behind the first four -- 4 up to n = 16 (54 registers: the kernel fits beside three
// window waves per SIMD), 16 for the long windows
template <int BATCH = 4>
__device__ __forceinline__ void dy_sparse_items(const ActWord* __restrict__ act_ring, u64* __restrict__ dst_bits,
                                                const unsigned* __restrict__ nzlist, unsigned total, int* __restrict__ rowcount,
                                                int* __restrict__ litcount, int* __restrict__ bbox, const PassGeom& g, int use_dy,
                                                unsigned first_e, unsigned ngroups) {
    const int words = g.H * g.WW;
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15, gbase = lane & 48;
    const int dy = sub / 3 - 1, k = sub % 3 - 1;  // sub >= 9: idle
    // all lanes of a wave run the same number of iterations (shuffles inside)
    for (unsigned e0 = first_e - (lane >> 4); e0 < total; e0 += ngroups) {
        const unsigned e = e0 + (lane >> 4);
        const bool live = e < total;
        const unsigned idx = live ? nzlist[e] : 0u;
        const int i = idx / words, gw = idx - i * words;
        const int y = gw / g.WW, w = gw - y * g.WW;
        const long long t = g.t_base + 1 + i;
        u64 out;
        if (use_dy) {
            const int len = window_len(g, i);
            const int yy = y + dy, ww = w + k;
            const bool row_in = sub < 9 && yy >= 0 && yy < g.H;
            const bool word_in = ww >= 0 && ww < g.WW;
            u64 all = ~0ull, first = 0;
            if (live && row_in) {
                if (word_in) {
                    // AND over the last `len` binaries, four at a time, until nothing is left of it: a moving streak's word is lit in
                    // two or three consecutive frames, so most lanes are done after the first four loads (round 3 issued all len
                    // of them -- 144 tagged 16-byte loads per listed word at n = 16 -- each behind a 64-bit modulo for its ring slot)
                    const int slot0 = (int)((t - 1) % g.R);  // ring slot of frame t; frame t - j sits j slots back (len <= n <= R)
                    const int cell = yy * g.WW + ww;
                    // (round 6: behind the first four, the rest in batches of SIXTEEN independent loads -- with windows of 25 / 30
                    // frames a persisting streak's word stays lit for most of them, and four at a time that was a chain of seven
                    // dependent round trips per listed word: 0.29 ms per 4000-frame pass at n = 25 against 0.10 at n = 8)
                    {
                        u64 v[4];
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) {
                            int sl = slot0 - k2;
                            if (sl < 0) sl += g.R;
                            v[k2] = k2 < len ? act_word(act_ring + (size_t)sl * words, cell, (u64)(t - k2)) : ~0ull;
                        }
                        first = v[0];
                        all &= v[0] & v[1] & v[2] & v[3];
                    }
                    for (int j0 = 4; j0 < len && all != 0; j0 += BATCH) {
                        u64 v[BATCH];
#pragma unroll
                        for (int k2 = 0; k2 < BATCH; ++k2) {
                            const int j = j0 + k2;
                            int sl = slot0 - j;
                            if (sl < 0) sl += g.R;
                            v[k2] = j < len ? act_word(act_ring + (size_t)sl * words, cell, (u64)(t - j)) : ~0ull;
                        }
                        u64 a = ~0ull;
#pragma unroll
                        for (int k2 = 0; k2 < BATCH; ++k2) a &= v[k2];
                        all &= a;
                    }
                } else {
                    all = 0;
                }
            }
            const u64 l = shfl64(all, gbase + ((sub + 15) & 15)), r = shfl64(all, gbase + ((sub + 1) & 15));
            u64 m = ~0ull;
            if (row_in) {
                const Win q = make_win(w, g.W);
                const u128 a128 = (u128)(l >> 61) | ((u128)all << 3) | ((u128)r << 67);
                m = (u64)((hor_and3_open(~a128 & q.valid, q) & q.valid) >> 3);
            }
            out = shfl64(first, gbase + 4) & shfl64(m, gbase + 1) & shfl64(m, gbase + 4) & shfl64(m, gbase + 7);
        } else {
            out = (live && sub == 0) ? act_word(act_ring + (size_t)((t - 1) % g.R) * words, gw, (u64)t) : 0ull;
        }
        if (live && sub == 0 && out) {
            dst_bits[(size_t)i * words + gw] = out;
            const int c = __popcll(out);
            atomicAdd(&rowcount[i * g.H + y], c);
            atomicAdd(&litcount[i], c);
            // bounding box of the lit pixels of the frame, kept as four running maxima so that a zero fill is the
            // neutral start: {W-1 - min x, max x, H-1 - min y, max y}
            atomicMax(&bbox[4 * i + 0], g.W - 1 - (w * 64 + __builtin_ctzll(out)));
            atomicMax(&bbox[4 * i + 1], w * 64 + 63 - __builtin_clzll(out));
            atomicMax(&bbox[4 * i + 2], g.H - 1 - y);
            atomicMax(&bbox[4 * i + 3], y);
        }
    }
}
template <int BATCH = 4>
__global__ void __launch_bounds__(256) k_dy_sparse(const ActWord* __restrict__ act_ring, u64* __restrict__ dst_bits,
                                                   const unsigned* __restrict__ nzlist, const unsigned* __restrict__ nzcount,
                                                   int* __restrict__ rowcount, int* __restrict__ litcount,
                                                   int* __restrict__ bbox, PassGeom g, int use_dy) {
    RAISE_WAVE_PRIORITY();
    dy_sparse_items<BATCH>(act_ring, dst_bits, nzlist, *nzcount, rowcount, litcount, bbox, g, use_dy, (blockIdx.x * blockDim.x + threadIdx.x) >> 4,
                    gridDim.x * (blockDim.x >> 4));
}

// dst (per slot, [K][H][WW]) is written only where it is non-zero, on an all-zero background; this restores the
// background before the slot's next pass.  nzlist (from k_morph_close) names every word that can be non-zero.
__global__ void __launch_bounds__(256) k_sparse_clear(u64* __restrict__ dst_bits, const unsigned* __restrict__ nzlist,
                                                      const unsigned* __restrict__ nzcount) {
    RAISE_WAVE_PRIORITY();
    const unsigned total = *nzcount;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) dst_bits[nzlist[e]] = 0;
}

// Everything a pass needs zeroed before its kernels run, in ONE launch (each separate memset / kernel costs a trip through
// the command processor, and the noise chain that follows is on the critical path of the pass): the counters the pixel
// stages accumulate into, the block bit counts the slot's previous pass touched, the noise sums, and the dst words that pass left behind
// (listed in nzlist; nzcount itself is reset by k_noise_finalize, which runs next on the same stream).
__global__ void __launch_bounds__(256) k_pass_prepare(unsigned* __restrict__ counters, unsigned n_counters,
                                                      unsigned* __restrict__ blockcnt, unsigned per_frame, unsigned Kcap,
                                                      unsigned* __restrict__ sums32, unsigned n_sums32, u64* __restrict__ dst_bits,
                                                      const unsigned* __restrict__ nzlist, const unsigned* __restrict__ nzcount) {
    RAISE_WAVE_PRIORITY();
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (unsigned e = tid; e < n_counters / 4; e += nth) ((uint4*)counters)[e] = z;
    for (unsigned e = (n_counters & ~3u) + tid; e < n_counters; e += nth) counters[e] = 0;
    for (unsigned e = tid; e < n_sums32; e += nth) sums32[e] = 0;
    if (nzcount) {
        const unsigned total = *nzcount;
        for (unsigned e = tid; e < total; e += nth) dst_bits[nzlist[e]] = 0;
    }
    // the block bit counts the slot's previous pass touched (the set bits of note_block's bitmap), and the bitmap itself
    if (blockcnt != nullptr) {
        unsigned* bitmap = blockcnt + (size_t)Kcap * per_frame;
        const unsigned stride = block_bitmap_stride(per_frame);
        const size_t total = (size_t)Kcap * stride;
        for (size_t e = tid; e < total; e += nth) {
            unsigned v = bitmap[e];
            if (v == 0) continue;
            const unsigned i = (unsigned)(e / stride), d = (unsigned)(e - (size_t)i * stride);
            unsigned* f = blockcnt + (size_t)i * per_frame + 32u * d;
            do {
                f[__builtin_ctz(v)] = 0;
                v &= v - 1u;
            } while (v);
            bitmap[e] = 0;
        }
    }
}

// Result packing: the segments of a pass, [K][MAX_LINES][4] with nraw[i] valid rows each, are packed back to back
// (frame order) so that the host fetches them with ONE contiguous copy; off[i] = first packed segment of frame i,
// off[K] = total.  One workgroup: a scan over the frames, then a thread per frame moves its segments.
// The per-frame lit counts are copied next to the other per-frame results, so that everything the host wants from a pass
// sits in one contiguous block (one device-to-host copy).
__global__ void __launch_bounds__(1024) k_pack_lines(const int4* __restrict__ lines, const int* __restrict__ nraw, int K,
                                                     int max_lines, int4* __restrict__ packed, int* __restrict__ off,
                                                     const int* __restrict__ litcount, int* __restrict__ litcopy,
                                                     const unsigned* __restrict__ nzwords, int* __restrict__ tier_count) {
    RAISE_WAVE_PRIORITY();
    for (int i = threadIdx.x; i < K; i += blockDim.x) litcopy[i] = litcount[i];
    if (threadIdx.x < 2) off[K + 1 + threadIdx.x] = nzwords ? (int)nzwords[threadIdx.x] : 0;  // words of b with set bits (u64)
    // the routing counters of the Hough set: every launch that read them is done (this kernel runs behind them), so they are
    // left at zero for the set's next pass -- one memset command per pass less on the Hough stream
    if (tier_count && threadIdx.x < HOUGH_TIERS + 4) tier_count[threadIdx.x] = 0;  // (+ the work queues' heads: k_hough3_big, k_hough3 per launch)
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < K; base += 1024) {
        const int i = base + tid;
        const int cnt = i < K ? min(nraw[i], max_lines) : 0;
        int incl = cnt;
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry_s;
        for (int k = 0; k < wave; ++k) before += wsum[k];
        const int start = before + incl - cnt;
        if (i < K) {
            off[i] = start;
            for (int q = 0; q < cnt; ++q) packed[start + q] = lines[(size_t)i * max_lines + q];
        }
        __syncthreads();
        if (tid == 1023) carry_s = start + cnt;
        __syncthreads();
    }
    if (tid == 0) off[K] = carry_s;
}

// Row-band tiling: the owned rows [row0, row0 + rows) of each frame's dst plane -> exchange buffer, frame i
// at slot(i) = (i % nsplit) * Kp + i / nsplit, rows_stride rows per slot (the buffer is zero-filled beforehand).
__global__ void __launch_bounds__(256) k_export_rows(const u64* __restrict__ dst_bits, u64* __restrict__ out, int K, int H, int WW,
                                                     int row0, int rows, int rows_stride, int nsplit, int Kp) {
    const int i = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * WW) return;
    const int slot = (i % nsplit) * Kp + i / nsplit;
    out[(size_t)slot * rows_stride * WW + e] = dst_bits[((size_t)i * H + row0) * WW + e];
}

// The other end: frame i of a Hough context is assembled from nbands pieces (piece b = rows [b*band_rows, ...) at
// ((b * K + i) * band_rows) * WW), together with what the Hough stage wants to know about it: per-row and per-frame lit
// counts and the bounding box of the lit pixels (as k_dy_sparse leaves them).
// Row bands of one process, one shared Hough context: the band's owned rows straight into that context's whole-frame planes
// (plane i, rows full_row0 ...), and the row / frame lit counts and the bounding box of the words that are not zero straight into
// its counters -- what k_assemble_planes would otherwise take from a second sweep over the assembled planes (2 GB per 500 8K
// frames, 0.58 ms at the end of every pass, behind the last band's export).
__global__ void __launch_bounds__(256) k_export_rows_count(const u64* __restrict__ dst_bits, u64* __restrict__ planes, int H, int WW,
                                                           int row0, int rows, int Hf, int W, int full_row0, int* __restrict__ rowcount,
                                                           int* __restrict__ litcount, int* __restrict__ bbox) {
    const int i = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * WW) return;
    const u64 v = dst_bits[((size_t)i * H + row0) * WW + e];
    planes[((size_t)i * Hf + full_row0) * WW + e] = v;
    if (v) {
        const int y = full_row0 + e / WW, w = e % WW;
        const int c = __popcll(v);
        atomicAdd(&rowcount[i * Hf + y], c);
        atomicAdd(&litcount[i], c);
        atomicMax(&bbox[4 * i + 0], W - 1 - (w * 64 + __builtin_ctzll(v)));
        atomicMax(&bbox[4 * i + 1], w * 64 + 63 - __builtin_clzll(v));
        atomicMax(&bbox[4 * i + 2], Hf - 1 - y);
        atomicMax(&bbox[4 * i + 3], y);
    }
}
__global__ void __launch_bounds__(256) k_assemble_planes(const u64* __restrict__ pieces, u64* __restrict__ dst_bits,
                                                         int* __restrict__ rowcount, int* __restrict__ litcount,
                                                         int* __restrict__ bbox, int K, int H, int W, int WW, int band_rows, int inplace) {
    const int i = blockIdx.y;
    const int gw = blockIdx.x * blockDim.x + threadIdx.x;
    if (gw >= H * WW) return;
    const int y = gw / WW, w = gw - y * WW;
    const int b = y / band_rows;
    u64 v = pieces[(((size_t)b * K + i) * band_rows + (y - b * band_rows)) * WW + w];
    if (w == WW - 1 && (W & 63)) v &= (1ull << (W & 63)) - 1;  // padding bits stay 0 whatever the sender left there
    if (!inplace) dst_bits[(size_t)i * H * WW + gw] = v;  // (inplace: the bands' export kernels wrote the planes where they are read)
    if (v) {
        const int c = __popcll(v);
        atomicAdd(&rowcount[i * H + y], c);
        atomicAdd(&litcount[i], c);
        atomicMax(&bbox[4 * i + 0], W - 1 - (w * 64 + __builtin_ctzll(v)));
        atomicMax(&bbox[4 * i + 1], w * 64 + 63 - __builtin_clzll(v));
        atomicMax(&bbox[4 * i + 2], H - 1 - y);
        atomicMax(&bbox[4 * i + 3], y);
    }
}

// bit image -> u8 {0,255} image (mdh_get_binary)
__global__ void k_expand_bits(const u64* __restrict__ bits, uint8_t* __restrict__ out, int H, int W, int WW) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    out[p] = ((bits[(size_t)y * WW + (x >> 6)] >> (x & 63)) & 1) ? 255 : 0;
}
__global__ void k_expand_act(const ActWord* __restrict__ row, uint8_t* __restrict__ out, int H, int W, int WW, u64 t) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int y = p / W, x = p - y * W;
    out[p] = ((act_word(row, y * WW + (x >> 6), t) >> (x & 63)) & 1) ? 255 : 0;
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

// Time-blocked variant for frames whose width is a multiple of 64 and whose bases are 16-byte aligned (the layout of
// k_stack_threshold_wide: a wave owns 16 words = 1024 pixels, a lane 16 consecutive pixels, one 128-bit buffer load per lane
// per frame) -- a wave walks through the frames of its time chunk with the three previous frames in registers, so every frame
// byte is fetched once per pass instead of four times, by 16x fewer load instructions.  |a - b| > T on packed u16 pairs:
// min(a - b, b - a) mod 2^16 is the absolute difference, and bit 15 of T - |d| is the comparison.  N = ring size (3 or 4:
// int(4 / fps * fps) can truncate to 3, and numpy's negative ring indices then wrap -- Detector.py:258-261).
// SPARSE (round 6): both difference images of a frame are stored like the M3 window kernel's -- only the words with set bits, as
// {bits, frame number} in cplanes [K][2][H][WW] (a reader takes a word whose tag is not its frame for zero), and a word of the SECOND
// plane (|f3 - f2| > T: the one the output is made of, dst = dilate(second & ~dilate(first))) flags its 8-row block in the block
// bitmap.  On a sky the planes are empty but for stars and streaks: up to round 5 the kernel wrote two plain bit planes per frame
// (a quarter of its traffic, mixed into its reads) and the stencil read every word of them back -- 0.37 of the peak at the pipeline's
// pace; now the kernel only reads, and the stencil (k_classic_close_t) visits the flagged blocks' neighbourhoods.
template <int N, bool SPARSE = false>
__global__ void __launch_bounds__(256) k_classic_diff_tb(const uint8_t* const* __restrict__ ptrtab, const int* __restrict__ thr_tab,
                                                         u64* __restrict__ planes, PassGeom g, int Tc,
                                                         ActWord* __restrict__ cplanes = nullptr, unsigned* __restrict__ blockcnt = nullptr) {
    constexpr int PPL = 16, RW = 8, DW = 4;
    typedef unsigned dvec __attribute__((ext_vector_type(DW)));
    const int lane = threadIdx.x & 63;
    const int words = g.H * g.WW;
    const int gw0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PPL;
    if (gw0 >= words) return;
    const int i0 = blockIdx.y * Tc;
    const int steps = min(Tc, g.K - i0);
    if (steps <= 0) return;
    const int q = lane & 3;
    const int gw = gw0 + (lane >> 2);
    const int frame_bytes = g.H * g.W;
    const int off = gw < words ? gw * 64 + q * PPL : 0x7FFFFFF0;  // past the frame -> zeros
    union PK {
        unsigned u;
        u16x2 v;
    };
    auto fetch = [&](int e) -> dvec {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)ptrtab[e], 0, frame_bytes, 0x00020000);
        return __builtin_bit_cast(dvec, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, WIDE_AUX));  // (read once per launch: nt, like the M3 window kernel)
    };
    auto unpack = [&](const dvec& d, PK (&p)[RW]) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const unsigned sel = 0x0C000C00u | (unsigned)(r & 3) | ((unsigned)(4 + (r & 3)) << 16);
            p[r].u = __builtin_amdgcn_perm(d[r / 4 + DW / 2], d[r / 4], sel);
        }
    };
    // hist[k] = frame (current - 1 - k); table entry of pass frame i is i + n - 1, so the prologue reads entries i0 + n - 2 - k
    PK hist[3][RW];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < N - 1) unpack(fetch(i0 + N - 2 - k), hist[k]);
        else {
#pragma unroll
            for (int r = 0; r < RW; ++r) hist[k][r].u = 0;
        }
    }
    dvec nxt = fetch(i0 + N - 1);
    for (int s = 0; s < steps; ++s) {
        const int i = i0 + s;
        const dvec cur = nxt;
        if (s + 1 < steps) nxt = fetch(i + N);
        PK f0[RW];
        unpack(cur, f0);
        const unsigned short T = (unsigned short)min(__builtin_amdgcn_readfirstlane(thr_tab[i]), 256);
        const u16x2 Tv = {T, T};
        unsigned acc1 = 0, acc2 = 0;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            // f_k = ring[cur - k] with numpy wrap: k % N frames back
            const u16x2 a0 = f0[r].v;
            const u16x2 a1 = (1 % N) == 0 ? a0 : hist[(1 % N) - 1][r].v;
            const u16x2 a2 = (2 % N) == 0 ? a0 : hist[(2 % N) - 1][r].v;
            const u16x2 a3 = (3 % N) == 0 ? a0 : hist[(3 % N) - 1][r].v;
            PK d1, d2;
            d1.v = Tv - __builtin_elementwise_min((u16x2)(a1 - a0), (u16x2)(a0 - a1));
            d2.v = Tv - __builtin_elementwise_min((u16x2)(a3 - a2), (u16x2)(a2 - a3));
            acc1 = (acc1 >> 1) | (d1.u & 0x80008000u);
            acc2 = (acc2 >> 1) | (d2.u & 0x80008000u);
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            hist[2][r] = hist[1][r];
            hist[1][r] = hist[0][r];
            hist[0][r] = f0[r];
        }
        // low half bits 8..15 = pixels 0..7, high half bits 24..31 = pixels 8..15 (see k_stack_threshold_wide)
        const unsigned bits1 = ((acc1 >> 8) & 0xFFu) | ((acc1 >> 16) & 0xFF00u);
        const unsigned bits2 = ((acc2 >> 8) & 0xFFu) | ((acc2 >> 16) & 0xFF00u);
        const unsigned both = bits1 | (bits2 << 16);
        if constexpr (SPARSE) {
            if (__ballot(both != 0) == 0) continue;  // (most wave-frames of a sky: nothing over the threshold in 1024 pixels)
        }
        // gather the four lanes of a word: lane 0 <- 1, 2 <- 3, then 0 <- 2 (quad-permute DPP)
        const unsigned odd = (unsigned)__builtin_amdgcn_mov_dpp((int)both, 0xF5, 0xF, 0xF, true);
        const unsigned p1 = (both & 0xFFFFu) | (odd << 16);          // plane 1, pixels 0..31 of the lane pair
        const unsigned p2 = (both >> 16) | (odd & 0xFFFF0000u);      // plane 2
        const unsigned p1u = (unsigned)__builtin_amdgcn_mov_dpp((int)p1, 0xAA, 0xF, 0xF, true);
        const unsigned p2u = (unsigned)__builtin_amdgcn_mov_dpp((int)p2, 0xAA, 0xF, 0xF, true);
        if constexpr (SPARSE) {
            const u64 w1 = (u64)p1 | ((u64)p1u << 32), w2 = (u64)p2 | ((u64)p2u << 32);
            if (q == 0 && gw < words && (w1 | w2)) {  // rare: stores and non-returning atomics only (see the M3 window kernel)
                int gwr = gw;
                asm volatile("" : "+v"(gwr));
                const u64 tag = (u64)(g.t_base + 1 + i);
                if (w1) cplanes[((size_t)i * 2 + 0) * words + gwr] = ActWord{w1, tag};
                if (w2) {
                    cplanes[((size_t)i * 2 + 1) * words + gwr] = ActWord{w2, tag};
                    note_block<true>(blockcnt, g, i, gwr, (unsigned)__popcll(w2));
                }
            }
        } else
        if (q == 0 && gw < words) {
            planes[((size_t)i * 2 + 0) * words + gw] = (u64)p1 | ((u64)p1u << 32);
            planes[((size_t)i * 2 + 1) * words + gw] = (u64)p2 | ((u64)p2u << 32);
        }
    }
}

// The ring of FOUR frames in the sparse format, rebuilt around what the M3 window kernel of round 6 does (the kernel above with
// SPARSE = true reached 0.53 of the peak: two packed absolute differences per pixel pair and frame, 24 register moves to age its
// three frames of history, one load in flight beyond the current frame, a descriptor and two table look-ups per frame on the
// scalar unit):
//  * the second difference of frame t, |f(t-3) - f(t-2)| > T(t), IS the first difference of frame t - 2 whenever T(t) = T(t-2)
//    (the threshold moves at most every interval * n frames): its bits are kept from two frames ago -- one register -- instead of
//    being recomputed; only the previous frame's pixels stay in registers.  When the threshold did change, and for the first two
//    frames of a time chunk, the two old frames are fetched again through the table (a wave-uniform rare path);
//  * bursts of four frames, double-buffered; pass tables (padded pointer table, global loads with a scalar base, the threshold
//    from a WinRec made with len = 1: the low half IS T); a time chunk is a whole number of bursts.
__global__ void __launch_bounds__(256) k_classic_diff4_sparse(const uint8_t* const* __restrict__ ptrtab, const WinRec* __restrict__ wrec,
                                                              ActWord* __restrict__ cplanes, unsigned* __restrict__ blockcnt, PassGeom g, int Tc) {
    constexpr int PPL = 16, RW = 8, DW = 4, U = 4;
    typedef unsigned dvec __attribute__((ext_vector_type(DW)));
    typedef const __attribute__((address_space(1))) dvec* gdvec_p;
    typedef const __attribute__((address_space(1))) uint8_t* gbyte_p;
    const int lane = threadIdx.x & 63;
    const int words = g.H * g.WW;
    const int gw0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PPL;
    if (gw0 >= words) return;
    const int i0 = blockIdx.y * Tc;
    const int steps = min(Tc, g.K - i0);
    if (steps <= 0) return;
    const int q = lane & 3;
    const int gw = gw0 + (lane >> 2);
    const int off = gw < words ? gw * 64 + q * PPL : 0;  // (past the frame: any valid address; such lanes store nothing)
    union PK {
        unsigned u;
        u16x2 v;
    };
    auto fetchp = [&](const uint8_t* base) -> dvec {
        unsigned o = (unsigned)off;
        asm volatile("" : "+v"(o));
        return __builtin_nontemporal_load((gdvec_p)((gbyte_p)base + o));
    };
    auto unpack = [&](const dvec& d, PK (&p)[RW]) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const unsigned sel = 0x0C000C00u | (unsigned)(r & 3) | ((unsigned)(4 + (r & 3)) << 16);
            p[r].u = __builtin_amdgcn_perm(d[r / 4 + DW / 2], d[r / 4], sel);
        }
    };
    // |a - b| > T on packed pairs: min(a - b, b - a) mod 2^16 is the absolute difference, bit 15 of T - |d| the comparison;
    // 16 result bits per lane: low half bits 8..15 = pixels 0..7, high half bits 24..31 = pixels 8..15, gathered below
    auto diff_bits = [&](const PK (&a)[RW], const PK (&b)[RW], unsigned T) -> unsigned {
        const u16x2 Tv = {(unsigned short)T, (unsigned short)T};
        unsigned acc = 0;
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            PK d;
            d.v = Tv - __builtin_elementwise_min((u16x2)(a[r].v - b[r].v), (u16x2)(b[r].v - a[r].v));
            acc = (acc >> 1) | (d.u & 0x80008000u);
        }
        return ((acc >> 8) & 0xFFu) | ((acc >> 16) & 0xFF00u);
    };
    // table entry of the NEWEST frame of pass frame i is i + 3; pt[s] = entry of step s of the current burst pair
    const uint8_t* const* pt = ptrtab + i0 + 3;
    const WinRec* wr = wrec + i0;
    PK prev[RW];
    unpack(fetchp(pt[-1]), prev);
    dvec buf[2][U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[0][u] = fetchp(pt[u]);
    unsigned b1_m1 = 0, b1_m2 = 0;       // first-difference bits of the previous frame and of the one before
    unsigned T_m1 = ~0u, T_m2 = ~0u;     // their thresholds (~0: not computed yet -- the first two steps of a chunk take the rare path)
    for (int sb = 0; sb < steps; sb += 2 * U) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (sb + h * U >= steps) break;
            dvec(&cur)[U] = buf[h];
            dvec(&nxt)[U] = buf[h ^ 1];
#pragma unroll
            for (int u = 0; u < U; ++u) nxt[u] = fetchp(pt[(h + 1) * U + u]);  // (padded table: past the pass the last frame again)
            unsigned Tq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) Tq[u] = wr[h * U + u].v & 0xFFFFu;  // (padding records: T = 256, nothing fires)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int s = sb + h * U + u;
                const int i = i0 + s;
                const unsigned T = Tq[u];
                PK f0[RW];
                unpack(cur[u], f0);
                const unsigned bits1 = diff_bits(prev, f0, T);
                unsigned bits2;
                if (T == T_m2) {
                    bits2 = b1_m2;
                } else {  // rare (wave-uniform): the threshold moved, or the chunk has just begun -- frames t-2 and t-3 again, through the table
                    PK f2[RW], f3[RW];
                    unpack(fetchp(pt[h * U + u - 2]), f2);
                    unpack(fetchp(pt[h * U + u - 3]), f3);
                    bits2 = diff_bits(f3, f2, T);
                }
#pragma unroll
                for (int r = 0; r < RW; ++r) prev[r] = f0[r];
                b1_m2 = b1_m1; b1_m1 = bits1;
                T_m2 = T_m1; T_m1 = T;
                const unsigned both = bits1 | (bits2 << 16);
                if (__ballot(both != 0) == 0) continue;  // (most wave-frames of a sky)
                const unsigned odd = (unsigned)__builtin_amdgcn_mov_dpp((int)both, 0xF5, 0xF, 0xF, true);
                const unsigned p1 = (both & 0xFFFFu) | (odd << 16);
                const unsigned p2 = (both >> 16) | (odd & 0xFFFF0000u);
                const unsigned p1u = (unsigned)__builtin_amdgcn_mov_dpp((int)p1, 0xAA, 0xF, 0xF, true);
                const unsigned p2u = (unsigned)__builtin_amdgcn_mov_dpp((int)p2, 0xAA, 0xF, 0xF, true);
                const u64 w1 = (u64)p1 | ((u64)p1u << 32), w2 = (u64)p2 | ((u64)p2u << 32);
                if (q == 0 && gw < words && (w1 | w2) && s < steps) {  // rare: stores and non-returning atomics only
                    int gwr = gw;
                    asm volatile("" : "+v"(gwr));
                    const u64 tag = (u64)(g.t_base + 1 + i);
                    if (w1) cplanes[((size_t)i * 2 + 0) * words + gwr] = ActWord{w1, tag};
                    if (w2) {
                        cplanes[((size_t)i * 2 + 1) * words + gwr] = ActWord{w2, tag};
                        note_block<true>(blockcnt, g, i, gwr, (unsigned)__popcll(w2));
                    }
                }
            }
        }
        pt += 2 * U;
        wr += 2 * U;
    }
    // The kernel needs 62 registers: EIGHT waves per SIMD -- every wave slot and every register of the machine, so that each launch
    // of the post stages (and the next pass's noise chain) waits for a workgroup of this one to leave.  Held at 88 registers (five
    // waves, 72 registers free per SIMD) the launch alone is as fast and the pipeline 3.5-4 % faster (gpurun_out/r06_42;
    // A/B builds: -DCLASSIC4_PIN_V='"v63"').
    asm volatile("" ::: CLASSIC4_PIN_V);
}

__global__ void __launch_bounds__(256) k_classic_stencil(const u64* __restrict__ planes, u64* __restrict__ dst_bits,
                                                         int* __restrict__ rowcount, int* __restrict__ litcount,
                                                         int* __restrict__ bbox, PassGeom g) {
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
    if (out) {
        const int c = __popcll(out);
        atomicAdd(&rowcount[i * g.H + y], c);
        atomicAdd(&litcount[i], c);
        // bounding box of the lit pixels, as four running maxima: {W-1 - min x, max x, H-1 - min y, max y}
        atomicMax(&bbox[4 * i + 0], g.W - 1 - (w * 64 + __builtin_ctzll(out)));
        atomicMax(&bbox[4 * i + 1], w * 64 + 63 - __builtin_clzll(out));
        atomicMax(&bbox[4 * i + 2], g.H - 1 - y);
        atomicMax(&bbox[4 * i + 3], y);
    }
}

// The same stencil with one thread per (word column, band of MORPH_RB rows): the rows slide through three-deep register
// pipelines (dilated first difference, masked second difference), so a word of either plane is fetched once per band
// (+ 4 / 2 halo rows) instead of 15 / 3 times per output word.  Bands whose planes are empty produce nothing but zeros.
__global__ void __launch_bounds__(256) k_classic_stencil_rb(const u64* __restrict__ planes, u64* __restrict__ dst_bits,
                                                            int* __restrict__ rowcount, int* __restrict__ litcount,
                                                            int* __restrict__ bbox, PassGeom g) {
    const int i = blockIdx.y;
    const int words = g.H * g.WW;
    const int nb = (g.H + MORPH_RB - 1) / MORPH_RB;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= nb * g.WW) return;
    const int band = tid / g.WW, w = tid - band * g.WW;
    const int y0 = band * MORPH_RB;
    const long long t = g.t_base + 1 + i;
    const bool active = t >= g.n;  // Detector.py:263: fewer than stack_maxsize frames -> no detection
    const Win q = make_win(w, g.W);
    const u64* p1 = planes + ((size_t)i * 2 + 0) * words;
    const u64* p2 = planes + ((size_t)i * 2 + 1) * words;
    u64* out_plane = dst_bits + (size_t)i * words;
    u128 h1a = 0, h1b = 0;   // hor_or3 of the first-difference rows yy-2, yy-1
    u128 k_a = 0, k_b = 0;   // hor_or3 of the kept second-difference rows (yy-1)-2, (yy-1)-1
#pragma unroll
    for (int it = 0; it < MORPH_RB + 4; ++it) {
        const int yy = y0 - 2 + it;  // first-difference row entering the pipeline
        u128 h1c = 0;
        if (active && yy >= 0 && yy < g.H) h1c = hor_or3(load_win(p1 + (size_t)yy * g.WW, w, g.WW));
        const int yk = yy - 1;       // row whose dilated first difference is complete now
        u128 k_c = 0;
        if (active && yk >= 0 && yk < g.H) {
            const u128 d1 = h1a | h1b | h1c;
            k_c = hor_or3(load_win(p2 + (size_t)yk * g.WW, w, g.WW) & ~d1 & q.valid);
        }
        const int yo = yy - 2;       // output row complete now
        if (yo >= y0 && yo < g.H && yo < y0 + MORPH_RB) {
            const u64 out = (u64)(((k_a | k_b | k_c) & q.valid) >> 3);
            out_plane[(size_t)yo * g.WW + w] = out;
            if (out) {
                const int c = __popcll(out);
                atomicAdd(&rowcount[i * g.H + yo], c);
                atomicAdd(&litcount[i], c);
                atomicMax(&bbox[4 * i + 0], g.W - 1 - (w * 64 + __builtin_ctzll(out)));
                atomicMax(&bbox[4 * i + 1], w * 64 + 63 - __builtin_clzll(out));
                atomicMax(&bbox[4 * i + 2], g.H - 1 - yo);
                atomicMax(&bbox[4 * i + 3], yo);
            }
        }
        h1a = h1b; h1b = h1c;
        k_a = k_b; k_b = k_c;
    }
}

// The same stencil for the SPARSE planes of k_classic_diff_tb<N, true>: only the blocks k_morph_select_touched<false> lists -- the
// 3x3 block neighbourhoods of the blocks in which the second difference has a set bit -- are visited, one lane per row of a block
// (eight blocks per wave and step), tagged words read straight from the planes.  dst is all-zero elsewhere (the words a pass
// writes are listed in nzlist, and the slot's next k_pass_prepare zeroes them), so only non-zero words are stored.
// total_blocks: the pass's number of listed blocks (one atomic per frame), the host's cue to fall back to the dense kernels when a
// lowered threshold lets the noise through.
__global__ void __launch_bounds__(64) k_classic_close_t(const ActWord* __restrict__ cplanes, const unsigned* __restrict__ worklist,
                                                        const unsigned* __restrict__ workcount, u64* __restrict__ dst_bits,
                                                        int* __restrict__ rowcount, int* __restrict__ litcount, int* __restrict__ bbox,
                                                        unsigned* __restrict__ nzlist, unsigned* __restrict__ nzcount, PassGeom g, int wpf,
                                                        unsigned long long* __restrict__ total_blocks) {
    RAISE_WAVE_PRIORITY();
    const int lane = threadIdx.x, grp = lane >> 3, r = lane & 7;
    const unsigned per_frame = (unsigned)(((g.H + MORPH_RB - 1) / MORPH_RB) * g.WW);
    const size_t words = (size_t)g.H * g.WW;
    for (unsigned u = blockIdx.x; u < (unsigned)g.K * (unsigned)wpf; u += gridDim.x) {
        const unsigned i = u / (unsigned)wpf, sub = u - i * (unsigned)wpf;
        const unsigned cnt = workcount[(size_t)i * WORKCNT_STRIDE];
        if (cnt == 0) continue;
        if (lane == 0 && sub == 0 && total_blocks != nullptr) atomicAdd(total_blocks, (unsigned long long)cnt);
        const long long t = g.t_base + 1 + (long long)i;
        if (t < g.n) continue;  // Detector.py:263: fewer than stack_maxsize frames -> no detection
        const ActWord* p1 = cplanes + (size_t)i * 2 * words;  // |f1 - f0| > T
        const ActWord* p2 = p1 + words;                       // |f3 - f2| > T
        const unsigned* seg = worklist + (size_t)i * per_frame;
        for (unsigned e0 = 8 * sub; e0 < cnt; e0 += 8 * (unsigned)wpf) {
            const unsigned e = e0 + (unsigned)grp;
            if (e >= cnt) continue;
            const int blk = (int)(seg[e] - i * per_frame);
            const int band = blk / g.WW, w = blk - band * g.WW;
            const int y = band * MORPH_RB + r;
            if (y >= g.H) continue;
            const Win q = make_win(w, g.W);
            u128 acc = 0;
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= g.H) continue;
                const u128 src = load_win_act(p2 + (size_t)yy * g.WW, w, g.WW, (u64)t);
                if (src == 0) continue;  // (nothing to keep on this row: the first difference need not be looked at)
                u128 d1 = 0;
                for (int ey = -1; ey <= 1; ++ey) {
                    const int y2 = yy + ey;
                    if (y2 < 0 || y2 >= g.H) continue;
                    d1 |= hor_or3(load_win_act(p1 + (size_t)y2 * g.WW, w, g.WW, (u64)t));
                }
                acc |= hor_or3(src & ~d1 & q.valid);
            }
            const u64 out = (u64)((acc & q.valid) >> 3);
            if (out) {
                const unsigned gw = (unsigned)(y * g.WW + w);
                dst_bits[(size_t)i * words + gw] = out;
                nzlist[atomicAdd(&nzcount[0], 1u)] = (unsigned)(i * words) + gw;
                const int c = __popcll(out);
                atomicAdd(&rowcount[i * g.H + y], c);
                atomicAdd(&litcount[i], c);
                atomicMax(&bbox[4 * i + 0], g.W - 1 - (w * 64 + __builtin_ctzll(out)));
                atomicMax(&bbox[4 * i + 1], w * 64 + 63 - __builtin_clzll(out));
                atomicMax(&bbox[4 * i + 2], g.H - 1 - y);
                atomicMax(&bbox[4 * i + 3], y);
            }
        }
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
    int serial;            // 1 = one-point-at-a-time replay even when the list fits in LDS (A/B and parity testing)
    int no_lds_acc;        // 1 = keep the whole accumulator in global memory (A/B and parity testing)
    double mask_area;
    const u64* dst_bits;   // [K][H][WW] pristine
    u64* work_bits;        // [slots][H][WW] working mask of the frames that cannot keep it in LDS (all-zero between frames)
    const int* rowcount;   // [K][H]
    const int* litcount;   // [K]
    int* accum;            // [slots][180*numrho], zero between frames
    unsigned* nz_orig;     // [slots][nz_cap] row-major list (one-wave kernel; kept for the clean-up)
    unsigned* nz_work;     // [slots][nz_cap] working list when it does not fit in LDS
    long long nz_cap;
    int* out_lines;        // [K][MAX_LINES][4]
    int* out_nraw;         // [K]
    double* out_gap;       // [K]
    const float* trig;     // [180][2] (cos, sin) as float32 of the float64 value
    long long* dbg;        // optional [K][8] cycle stamps (MDH_HOUGH_DEBUG=1), else null
    long long* dbg_phase;  // optional [K][8] phase cycle totals of the four-wave kernel
    int list_cap;          // one-wave kernel: LDS point-list capacity (dwords)
    int cell_cap;          // one-wave kernel: LDS accumulator capacity (16-bit cells, even)
    // frame routing (k_hough_classify): every launch walks the frame list of its tier
    const int* flist;      // frames of this launch's tier
    const int* fcount;     // their number
    int* tier_list;        // [HOUGH_TIERS][K] lists being built by the classifier
    int* tier_count;       // [HOUGH_TIERS] + the heads of the launches' work queues
    int* fqueue;           // this launch's queue head (k_hough3; nullptr: the frames are dealt out by workgroup index, up to round 5)
    int K;                 // frames in the pass
    int lds_small, lds_large;  // dynamic LDS bytes of the two three-wave launches (the classifier's fit test)
    int lds_tiny;          // ... and of the launch for frames that need even less (tier 4; 0: no such launch)
    int all_one_wave;      // MDH_HOUGH_ONE_WAVE / MDH_HOUGH_SERIAL: every frame goes to the one-wave kernel
    int lds_bytes;         // dynamic LDS bytes of this launch
    const unsigned short* perm;  // visiting-order tables: the table of an N-point frame starts at N(N-1)/2 (N <= HOUGH_MAX_LIST)
    const unsigned* rnd;         // the first HOUGH_BIG_LIST outputs of cv::RNG(-1): a larger frame builds its table itself ...
    unsigned short* perm_big;    // ... into its pool slot's [HOUGH_BIG_LIST] entries of this buffer (hough3_frame<true>)
    const int* bbox;       // [K][4] {W-1 - min x, max x, H-1 - min y, max y} of the lit pixels
    int mute;              // the first `mute` frames of the pass are not replayed (time-window warm-up: results dropped anyway)
    int band_routing;      // 1 = frames that miss the small launch by their corner windows are routed by their band windows (round 6; MDH_HOUGH_BANDS=0: off)
    int acc_mode;          // 1 = full-accumulator Hough (h3_accumulate) instead of the exact replay, in the three-wave kernels
};


// Maximum over the 64 lanes of a wave (all lanes active), returned wave-uniform: four row_shr steps leave every 16-lane row's
// maximum in its lane 15, row_bcast:15 / row_bcast:31 carry it on to lane 63 -- seven DPP moves.  (__builtin_amdgcn_wave_reduce_max_u32
// compiles to a scalar loop over the 64 lanes on gfx950, ~2000 cycles: the largest single item of a PPHT trigger, round 3.)
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false));  // row_shr:1
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false));  // row_shr:2
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false));  // row_shr:4
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false));  // row_shr:8
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false));  // row_bcast:15 into rows 1 and 3
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false));  // row_bcast:31 into rows 2 and 3
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

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

// Per-wave context of one frame's PPHT.
struct Ppht {
    const HoughParams& P;
    u64* work;
    unsigned* work32;
    int* acc;
    float tc[3], ts[3];  // this lane's 3 angles
    int line_gap, frame, found, lane;
    int groups, lines;  // statistics (debug)
    long long t_shuffle, t_mask, t_vote, t_key, t_line;  // cycle totals per section (only with MDH_HOUGH_DEBUG)
    bool timing;
    // LDS-resident part of the accumulator: for angle n the rho bins [lo, hi] live in 16-bit LDS cells starting at
    // `off` (two cells per dword, each biased by CELL_BIAS so that +-1 on one half never carries into the other);
    // bins outside the window stay in the global accumulator.
    unsigned* cells;
    int dummy_base;  // 64 spare dwords behind the cells: one private dummy target per lane
    int lo[3], hi[3], off[3], span[3];  // span = hi - lo (-1: no window for this angle)
    bool touched_global;
};
#define CELL_BIAS 0x4000

__device__ __forceinline__ bool work_bit(const Ppht& C, int x, int y) {
    return (ld_l2(&C.work[(size_t)y * C.P.WW + (x >> 6)]) >> (x & 63)) & 1;
}

// ++accum[n][rho(x,y)] for this lane's angles; key = ((new value + bias) << 8) | (255 - n): the wave-wide maximum of
// the key is the FIRST maximum of the reference's scan over n.  Counts can be negative (OpenCV takes votes back for
// line pixels that never voted), hence the bias.
#define KEY_BIAS (1 << 22)
__device__ __forceinline__ int key_value(unsigned key) { return (int)(key >> 8) - KEY_BIAS; }
// Branch-free in the common case: every lane issues its LDS atomic (lanes without a valid angle or with a bin
// outside the window hit a private dummy cell), and the global-accumulator path is entered only when some lane
// really is outside its window (wave-uniform test).
__device__ __forceinline__ void vote_issue(Ppht& C, int x, int y, int (&old)[3]) {
    const float fx = (float)x, fy = (float)y;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int n = C.lane + 64 * q;
        const int r = __float2int_rn(__fadd_rn(__fmul_rn(fx, C.tc[q]), __fmul_rn(fy, C.ts[q]))) + C.P.half;
        const unsigned d = (unsigned)(r - C.lo[q]);
        const bool inwin = C.span[q] >= 0 && d <= (unsigned)C.span[q];  // span = hi - lo, or -1 for "no window / no angle"
        const int c = C.off[q] + (int)d;
        const int sh = (c & 1) * 16;
        const unsigned o = atomicAdd(&C.cells[inwin ? (c >> 1) : C.dummy_base + C.lane], inwin ? (1u << sh) : 0u);
        old[q] = inwin ? (int)((o >> sh) & 0xFFFFu) - CELL_BIAS : -KEY_BIAS - 1;
        const bool outside = !inwin && n < NUM_ANGLE;
        if (__any(outside)) {
            if (outside) {
                old[q] = atomicAdd(&C.acc[(size_t)n * C.P.numrho + r], 1);
                C.touched_global = true;
            }
        }
    }
}
__device__ __forceinline__ unsigned vote_key(const Ppht& C, const int (&old)[3]) {
    unsigned key = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const unsigned k = ((unsigned)(old[q] + 1 + KEY_BIAS) << 8) | (unsigned)(255 - (C.lane + 64 * q));
        key = k > key ? k : key;
    }
    return key;
}
__device__ __forceinline__ void unvote(Ppht& C, int x, int y) {
    const float fx = (float)x, fy = (float)y;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int n = C.lane + 64 * q;
        const int r = __float2int_rn(__fadd_rn(__fmul_rn(fx, C.tc[q]), __fmul_rn(fy, C.ts[q]))) + C.P.half;
        const unsigned d = (unsigned)(r - C.lo[q]);
        const bool inwin = C.span[q] >= 0 && d <= (unsigned)C.span[q];
        const int c = C.off[q] + (int)d;
        atomicSub(&C.cells[inwin ? (c >> 1) : C.dummy_base + C.lane], inwin ? (1u << ((c & 1) * 16)) : 0u);
        const bool outside = !inwin && n < NUM_ANGLE;
        if (__any(outside)) {
            if (outside) {
                atomicSub(&C.acc[(size_t)n * C.P.numrho + r], 1);
                C.touched_global = true;
            }
        }
    }
}

// The point (x, y) reached the threshold at angle index bn: extract the segment, clear its pixels from the
// working mask, take their votes back if the segment is long enough, emit it (hough.cpp, second half of the loop).
__device__ void ppht_line(Ppht& C, int x, int y, int bn) {
    ++C.lines;
    const HoughParams& P = C.P;
    const int lane = C.lane;
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
            if (inside) lit = work_bit(C, j1, i1);
            const u64 mi = __ballot(inside), ml = __ballot(lit);
            // scan the 64 steps run by run (lit runs reset the gap, dark runs extend it) instead of bit by bit
            const int n_in = mi == ~0ull ? 64 : __builtin_ctzll(~mi);  // steps before the image border
            int pos = 0;
            while (pos < n_in) {
                const u64 rest = ml >> pos;
                if (rest & 1) {
                    const u64 inv = ~rest;
                    int run = inv ? __builtin_ctzll(inv) : 64;
                    run = min(run, n_in - pos);
                    pos += run;
                    last_lit = s0 + pos - 1;
                    gap = 0;
                } else {
                    int run = rest ? __builtin_ctzll(rest) : 64;
                    run = min(run, n_in - pos);
                    if (gap + run > C.line_gap) { done = true; break; }
                    gap += run;
                    pos += run;
                }
            }
            if (!done && n_in < 64) done = true;  // left the image
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
                const unsigned old = atomicAnd(&C.work32[((size_t)i1 * P.WW + (j1 >> 6)) * 2 + ((j1 >> 5) & 1)], ~bit);
                was = old & bit;
            }
            u64 ms = __ballot(was);
            if (good) {
                while (ms) {
                    const int bb = __builtin_ctzll(ms);
                    ms &= ms - 1;
                    const int sb = s0 + bb;
                    const int X = x0 + dx * sb, Y = y0 + dy * sb;
                    unvote(C, xflag ? X : X >> 16, xflag ? Y >> 16 : Y);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (good) {
        if (C.found < MAX_LINES && lane == 0) {
            int* o = P.out_lines + ((size_t)C.frame * MAX_LINES + C.found) * 4;
            o[0] = end_x[0]; o[1] = end_y[0]; o[2] = end_x[1]; o[3] = end_y[1];
        }
        ++C.found;
    }
}

// minimum over lanes 0..15 (one DPP row), broadcast: 4 row_shr steps leave it in lane 15
__device__ __forceinline__ unsigned row16_min_lane15(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x111, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x112, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x114, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x118, 0xF, 0xF, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}
__device__ __forceinline__ u64 rng_step(u64 s) { return (u64)(unsigned)s * 4164903690ull + (s >> 32); }

// Serial replay (any list size; the list lives in global memory when it does not fit in LDS).
template <bool LDS_LIST>
__device__ void ppht_serial(Ppht& C, int N, unsigned* list) {
    u64 state = 0xFFFFFFFFFFFFFFFFull;
    for (int count = N; count > 0; --count) {
        state = rng_step(state);
        const int idx = (int)((unsigned)state % (unsigned)count);
        unsigned pt;
        if (LDS_LIST) {
            pt = list[idx];
            const unsigned last = list[count - 1];
            if (C.lane == 0) list[idx] = last;
        } else {
            pt = ld_l2(&list[idx]);
            const unsigned last = ld_l2(&list[count - 1]);
            if (C.lane == 0) st_l2(&list[idx], last);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const int x = pt & 0xFFFF, y = pt >> 16;
        if (!work_bit(C, x, y)) continue;
        int old[3];
        vote_issue(C, x, y, old);
        const unsigned key = vote_key(C, old);
        if (!__any(key_value(key) >= C.P.threshold)) continue;
        const unsigned best = wave_max_u32(key);
        ppht_line(C, x, y, 255 - (int)(best & 0xFF));
    }
}

// Group-speculative replay (list in LDS).  The visiting order depends only on the RNG and the list length, never
// on pixel data, so PPHT_G consecutive draws are resolved at once: lane k takes draw k (index = rnd_k % (count-k)),
// the swap-removes are applied together when they cannot interact (no index inside the tail window, no repeated
// index; otherwise one ordinary serial draw is made), the working-mask bits of the G points are fetched in one
// round trip and the votes of all live points are issued back to back (same-address atomics complete in issue
// order, so each point's returned counts are exactly the sequential ones).  If some point reaches the threshold,
// the votes of the points after it are taken back, its line is processed, and the rest of the group is re-examined
// against the updated mask -- the result is identical to the one-point-at-a-time loop.
#define PPHT_G 16
__device__ void ppht_grouped(Ppht& C, int N, unsigned* list) {
    const int lane = C.lane;
    u64 state = 0xFFFFFFFFFFFFFFFFull;
    int count = N;
    while (count > 0) {
        const long long ta = C.timing ? clock64() : 0;
        int gn = count < PPHT_G ? count : PPHT_G;
        u64 st = state, my_state = 0;
        unsigned my_rnd = 0;
#pragma unroll
        for (int k = 0; k < PPHT_G; ++k) {
            if (k < gn) {
                st = rng_step(st);
                if (lane == k) {
                    my_rnd = (unsigned)st;
                    my_state = st;
                }
            }
        }
        const bool in_grp = lane < gn;
        const int my_idx = in_grp ? (int)(my_rnd % (unsigned)(count - lane)) : -1 - lane;
        // Largest prefix g of the draws whose swap-removes cannot interact: every index must lie below the tail
        // window [count-g, count) (i.e. its distance from the end is >= g) and must not repeat an earlier one.
        int lim = in_grp ? count - 1 - my_idx : PPHT_G;
#pragma unroll
        for (int j = 0; j < PPHT_G - 1; ++j) {
            const int v = __builtin_amdgcn_readlane(my_idx, j);
            if (in_grp && j < lane && v == my_idx) lim = lane < lim ? lane : lim;
        }
        const int g = min(gn, (int)row16_min_lane15((unsigned)lim));
        unsigned my_pt = 0;
        if (g == 0) {  // the first draw removes the last element itself: ordinary serial step
            state = rng_step(state);
            const int idx = (int)((unsigned)state % (unsigned)count);
            my_pt = list[idx];
            const unsigned last = list[count - 1];
            if (lane == 0) list[idx] = last;
            gn = 1;
        } else {
            const bool mine = lane < g;
            unsigned tail = 0;
            if (mine) {
                my_pt = list[my_idx];
                tail = list[count - 1 - lane];
            }
            if (mine) list[my_idx] = tail;
            gn = g;
            {  // the generator state after the g-th draw lives in lane g-1 (g is wave-uniform: one v_readlane pair)
                const int src = __builtin_amdgcn_readfirstlane(g - 1);
                state = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(my_state >> 32), src) << 32) |
                        (unsigned)__builtin_amdgcn_readlane((int)(unsigned)my_state, src);
            }
        }
        count -= gn;
        ++C.groups;
        if (C.timing) C.t_shuffle += clock64() - ta;
        u64 pending = (1ull << gn) - 1;
        const int px = my_pt & 0xFFFF, py = my_pt >> 16;
        while (pending) {
            const long long tb = C.timing ? clock64() : 0;
            bool alive = false;
            if ((pending >> lane) & 1) alive = work_bit(C, px, py);
            const u64 am = __ballot(alive);
            if (C.timing) C.t_mask += clock64() - tb;
            if (!am) break;
            const long long tc0 = C.timing ? clock64() : 0;
            int old[PPHT_G][3];
#pragma unroll
            for (int k = 0; k < PPHT_G; ++k) {
                if ((am >> k) & 1) {
                    const unsigned pt = __builtin_amdgcn_readlane(my_pt, k);
                    vote_issue(C, pt & 0xFFFF, pt >> 16, old[k]);
                }
            }
            const long long td = C.timing ? clock64() : 0;
            if (C.timing) C.t_vote += td - tc0;
            int trig = -1;
            unsigned trig_key = 0;
#pragma unroll
            for (int k = 0; k < PPHT_G; ++k) {
                if (trig < 0 && ((am >> k) & 1)) {
                    const unsigned key = vote_key(C, old[k]);
                    if (__any(key_value(key) >= C.P.threshold)) {
                        trig = k;
                        trig_key = key;
                    }
                }
            }
            if (C.timing) C.t_key += clock64() - td;
            if (trig < 0) break;
            const long long te = C.timing ? clock64() : 0;
            // points after the trigger voted too early: take those votes back, they are re-examined below
            const u64 later = am & ~((2ull << trig) - 1);
#pragma unroll
            for (int k = 1; k < PPHT_G; ++k) {
                if ((later >> k) & 1) {
                    const unsigned pt = __builtin_amdgcn_readlane(my_pt, k);
                    unvote(C, pt & 0xFFFF, pt >> 16);
                }
            }
            const unsigned best = wave_max_u32(trig_key);
            unsigned tp = 0;
#pragma unroll
            for (int k = 0; k < PPHT_G; ++k)
                if (k == trig) tp = __builtin_amdgcn_readlane(my_pt, k);
            ppht_line(C, tp & 0xFFFF, tp >> 16, 255 - (int)(best & 0xFF));
            if (C.timing) C.t_line += clock64() - te;
            pending = later;
        }
    }
}

// One frame: `i` indexes the pass's per-frame arrays (planes, counts, results), `slot` the pooled global buffers (accumulator,
// working mask, point lists) -- a workgroup keeps its slot for every frame it walks through.
__device__ void ppht_one_wave_frame(const HoughParams& P, int i, int slot, unsigned* lds_dyn) {
    // dynamic LDS: [ point list: list_cap dwords | accumulator cells: cell_cap/2 dwords | 64 dummy dwords ]
    // The split is PER FRAME (round 5).  Round 4 fixed it at 8192 list entries: a frame with more lit pixels -- a cloud edge, a
    // wavy strip of 12-17 k pixels across the whole frame -- kept its point list in GLOBAL memory and was replayed one point at
    // a time, three dependent L2 round trips per point (list swap, working-mask bit, returning votes): 17 ms per frame.  A list
    // of up to ~31 k points still fits next to 16 k accumulator cells; such a frame takes the group-speculative replay
    // (ppht_grouped: sixteen draws resolved at once, their mask bits in one round trip, their votes issued back to back) with
    // the smallest rho windows in LDS and the others in the pooled global accumulator.
    const int lane = threadIdx.x;
    const int N = P.litcount[i];
    const int lds_dwords = P.list_cap + P.cell_cap / 2;
    int list_cap = P.list_cap, cell_cap = P.cell_cap;
    if (N > list_cap && N <= lds_dwords - 8192) {
        list_cap = (N + 63) & ~63;
        cell_cap = 2 * (lds_dwords - list_cap);
    }
    unsigned* lds_list = lds_dyn;
    unsigned* lds_cells = lds_dyn + list_cap;
    const long long c0 = P.dbg ? clock64() : 0;
    const int line_gap = (int)rint(P.out_gap[i]);  // maxLineGap (k_hough_classify) -> cvRound
    // Accumulator windows: rho is monotone in x and in y for a fixed angle (rounded float products and sum), so over
    // the bounding box of the lit pixels its extremes are at the corners.  Window sizes decide which launch owns
    // the frame (both launches evaluate the same test) and whether the accumulator lives in LDS at all.
    int w_lo[3] = {0, 0, 0}, w_size[3] = {0, 0, 0}, w_span[3] = {-1, -1, -1}, w_off[3] = {0, 0, 0};
    int cells_needed = 0;
    {
        const int bx0 = P.W - 1 - P.bbox[4 * i], bx1 = P.bbox[4 * i + 1], by0 = P.H - 1 - P.bbox[4 * i + 2], by1 = P.bbox[4 * i + 3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int n = lane + 64 * q;
            if (n < NUM_ANGLE) {
                const float c = P.trig[2 * n], sn = P.trig[2 * n + 1];
                const int r00 = rho_bin(bx0, by0, c, sn, P.half), r10 = rho_bin(bx1, by0, c, sn, P.half);
                const int r01 = rho_bin(bx0, by1, c, sn, P.half), r11 = rho_bin(bx1, by1, c, sn, P.half);
                w_lo[q] = min(min(r00, r10), min(r01, r11));
                w_size[q] = max(max(r00, r10), max(r01, r11)) - w_lo[q] + 1;
            }
            cells_needed += w_size[q];
        }
        for (int o = 32; o > 0; o >>= 1) cells_needed += __shfl_xor(cells_needed, o);
    }
    // Which windows live in LDS: all of them if they fit; otherwise the smallest ones (the angles across the streak,
    // where the votes pile up), found by bisecting the size limit; the remaining angles use the global accumulator.
    {
        int limit = 0x7FFFFFFF;
        if (cells_needed > cell_cap) {
            int lo = 0, hi = P.numrho;  // largest `limit` with sum(size <= limit) <= cell_cap
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                int sum = 0;
#pragma unroll
                for (int q = 0; q < 3; ++q) sum += w_size[q] <= mid ? w_size[q] : 0;
                for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                if (sum <= cell_cap) lo = mid; else hi = mid - 1;
            }
            limit = lo;
        }
        int base = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int size = (w_size[q] > 0 && w_size[q] <= limit) ? w_size[q] : 0;
            int incl = size;
            for (int d = 1; d < 64; d <<= 1) {
                const int tt = __shfl_up(incl, d);
                if (lane >= d) incl += tt;
            }
            w_off[q] = base + incl - size;
            w_span[q] = size - 1;
            base += __shfl(incl, 63);
        }
        cells_needed = base;  // cells actually placed in LDS
    }
    const size_t words = (size_t)P.H * P.WW;
    const u64* dst = P.dst_bits + (size_t)i * words;
    u64* wrk = P.work_bits + (size_t)slot * words;
    unsigned* nzo = P.nz_orig + (size_t)slot * P.nz_cap;
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
                wrk[(size_t)y * P.WW + w] = word;  // working copy: only rows with lit pixels (the rest stays zero)
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
    const long long c1 = P.dbg ? clock64() : 0;
    Ppht C{P, wrk, (unsigned*)wrk,
           P.accum + (size_t)slot * NUM_ANGLE * P.numrho, {0, 0, 0}, {0, 0, 0}, line_gap, i, 0, lane, 0, 0, 0, 0, 0, 0, 0, P.dbg != nullptr,
           lds_cells, cell_cap / 2, {0, 0, 0}, {-1, -1, -1}, {0, 0, 0}, {-1, -1, -1}, false};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int n = lane + 64 * q;
        C.tc[q] = n < NUM_ANGLE ? P.trig[2 * n] : 0.f;
        C.ts[q] = n < NUM_ANGLE ? P.trig[2 * n + 1] : 0.f;
    }
    if (!P.no_lds_acc && N <= list_cap && cells_needed <= cell_cap) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            C.lo[q] = w_lo[q];
            C.span[q] = w_span[q];
            C.hi[q] = w_lo[q] + w_span[q];
            C.off[q] = w_off[q];
        }
        for (int p = lane; p < (cells_needed + 1) / 2; p += 64) lds_cells[p] = (CELL_BIAS << 16) | CELL_BIAS;
    }
    if (N <= list_cap) {
        for (int p = lane; p < N; p += 64) lds_list[p] = ld_l2(&nzo[p]);
        __threadfence_block();  // one wave: its LDS operations execute in order
        if (P.serial)
            ppht_serial<true>(C, N, lds_list);
        else
            ppht_grouped(C, N, lds_list);
    } else {
        unsigned* nzw = P.nz_work + (size_t)slot * P.nz_cap;
        for (int p = lane; p < N; p += 64) st_l2(&nzw[p], ld_l2(&nzo[p]));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ppht_serial<false>(C, N, nzw);
    }
    if (lane == 0) P.out_nraw[i] = C.found;
    const long long c2 = P.dbg ? clock64() : 0;
    // leave the global accumulator all-zero: every cell ever touched belongs to some point of the original list
    if (__any(C.touched_global)) {
        int* acc = P.accum + (size_t)slot * NUM_ANGLE * P.numrho;
        for (int p = 0; p < N; ++p) {
            const unsigned pt = ld_l2(&nzo[p]);
            const int x = pt & 0xFFFF, y = pt >> 16;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int n = lane + 64 * q;
                if (n < NUM_ANGLE) {
                    const int r = rho_bin(x, y, C.tc[q], C.ts[q], P.half);
                    if (C.span[q] < 0 || (unsigned)(r - C.lo[q]) > (unsigned)C.span[q]) acc[(size_t)n * P.numrho + r] = 0;
                }
            }
        }
    }
    // the working mask is all-zero between passes: clear the rows that were copied in
    for (int yb = 0; yb < P.H; yb += 64) {
        const int y = yb + lane;
        if (y < P.H && P.rowcount[i * P.H + y] > 0)
            for (int w = 0; w < P.WW; ++w) wrk[(size_t)y * P.WW + w] = 0;
    }
    if (P.dbg && lane == 0) {
        long long* d = P.dbg + (size_t)i * 8;
        d[0] = N; d[1] = c1 - c0; d[2] = c2 - c1; d[3] = C.t_shuffle; d[4] = C.t_mask; d[5] = C.t_vote; d[6] = C.t_key; d[7] = C.t_line;
    }
}

// Frames the classifier sent to the one-wave replay (more points than an LDS list holds, or every frame under
// MDH_HOUGH_ONE_WAVE / MDH_HOUGH_SERIAL): gridDim.x workgroups walk the tier's frame list, each with its own buffer slot.
__global__ void __launch_bounds__(64) k_hough_ppht(HoughParams P) {
    RAISE_WAVE_PRIORITY();
    extern __shared__ __attribute__((aligned(16))) unsigned lds_dyn[];
    const int count = *P.fcount;
    for (int j = blockIdx.x; j < count; j += gridDim.x) {
        ppht_one_wave_frame(P, P.flist[j], blockIdx.x, lds_dyn);
        __threadfence();
        __syncthreads();  // the next frame reuses the LDS and the slot's (clean again) global buffers
    }
}

// ================================================================================================
// Frame routing.  One wave per frame: the adaptive maxLineGap (Detector.py:342-344), and which launch replays the frame:
//   tier 0 / 1  three-wave kernel, everything in LDS (point list, working mask over the bounding box, all 180 rho windows),
//               small / large dynamic LDS (2-3 frames per CU / one frame per CU)
//   tier 4      the same kernel with 40 KB (four frames per CU): short streaks -- every frame at 960x540, where the Hough stage is
//               the largest item next to the window kernel (round 4: +3-5 % at the pipeline's pace there, nothing at 1080p)
//   tier 2      three-wave kernel, hybrid: the rho windows that fit in LDS + a pooled global accumulator for the rest
//   tier 3      one-wave kernel with a global point list (more than HOUGH_MAX_LIST points)
//   none        fewer points than the vote threshold: no accumulator cell can reach it, so there is no segment
#define HOUGH_MAX_LIST 8192
#define HOUGH_BIG_LIST 32768  // frames of up to this many points may take the three-wave replay with a visiting-order table of their own (if their LDS need allows)
#define HOUGH3_THREADS 192
#define HOUGH3_FIXED_DWORDS 64  // mailboxes + scan scratch + line results

// A line clears at most one pixel per step along its major axis, and only inside the bounding box of the lit pixels.
__device__ __forceinline__ int hough3_uv_cap(int N, int bw, int bh) { return (min(N, max(bw, bh) + 2) + 3) & ~3; }
__device__ __forceinline__ int hough3_lds_need(int N, int uv_cap, int cells, int mask_dwords) {
    // [fixed | point list N | un-vote list (two floats per entry) | mask | cells + one dummy dword per thread]
    return 4 * (HOUGH3_FIXED_DWORDS + ((N + 3) & ~3) + 2 * uv_cap + mask_dwords + (cells + 1) / 2 + HOUGH3_THREADS);
}

// One wave per frame.  Returns the tier (-1: nothing to replay).
// Round 6: BAND WINDOWS in the routing.  The rho windows of a frame's 180 angles come from the corners of the bounding box of its lit
// pixels; a streak is thin and oblique, so the range its points really span is 15-35 % smaller -- and at long windows (n = 25: a
// streak stays in the window for 37 of 50 frames, two or three streaks per frame) a quarter to a half of the frames need 85-100 KB by
// their corners: the whole LDS of a CU each (tier 1) instead of half of it.  A frame that misses the small launch by its corner
// windows is looked at again with the x-extent of its lit pixels per band of HB_ROWS rows (what hough3_frame derives from its point
// list when the corner windows do not fit: the same boxes, hence the same windows); if THOSE fit, the frame takes the small launch.
#ifndef HOUGH_BAND_ROUTING
#define HOUGH_BAND_ROUTING 1
#endif
#define HOUGH_HB_ROWS 8          // rows per band
#define HOUGH_CLS_MAXBANDS 136   // bands the routing looks at (bounding boxes of up to 1088 rows); scratch: 2 x this many dwords per wave
__device__ int hough_classify_frame(const HoughParams& P, int i, int lane, unsigned* scratch) {
    const int N = P.litcount[i];
    double gap = P.max_gap;
    if (!P.fixed_gap) {
        const double dst_sum = (double)N / P.mask_area * 100;
        double gg = 1 - dst_sum / 0.05;
        if (gg < 0) gg = 0;
        gap = gg * P.max_gap;
    }
    if (lane == 0) P.out_gap[i] = gap;
    if (N < P.threshold || i < P.mute) {  // (threshold >= 1, so this covers the empty frame)
        if (lane == 0) P.out_nraw[i] = 0;
        return -1;
    }
    const int bx0 = P.W - 1 - P.bbox[4 * i], bx1 = P.bbox[4 * i + 1], by0 = P.H - 1 - P.bbox[4 * i + 2], by1 = P.bbox[4 * i + 3];
    int cells = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int n = lane + 64 * q;
        int size = 0;
        if (n < NUM_ANGLE) {
            const float c = P.trig[2 * n], sn = P.trig[2 * n + 1];
            const int r00 = rho_bin(bx0, by0, c, sn, 0), r10 = rho_bin(bx1, by0, c, sn, 0);
            const int r01 = rho_bin(bx0, by1, c, sn, 0), r11 = rho_bin(bx1, by1, c, sn, 0);
            size = max(max(r00, r10), max(r01, r11)) - min(min(r00, r10), min(r01, r11)) + 1;
        }
        cells += max(size, __shfl_xor(size, 1));  // angles 2a and 2a + 1 share the dwords of the larger window (hough3_frame)
    }
    for (int o = 32; o > 0; o >>= 1) cells += __shfl_xor(cells, o);
    const int mask_dw = (by1 - by0 + 1) * ((bx1 >> 6) - (bx0 >> 6) + 1) * 2;
    if (P.all_one_wave) return 3;
    if (N > HOUGH_MAX_LIST) {
        // Round 5: more points than the shared visiting-order tables cover (a cloud edge: 12-17 k points).  Such a frame takes the
        // hybrid three-wave replay with a table of its own, built by the workgroup in the LDS its accumulator cells get afterwards
        // -- if the point list, the un-vote list and that scratch (N dwords) fit next to each other.  Otherwise (and in the
        // accumulator mode, whose specification replays such frames exactly in the one-wave kernel) the one-wave kernel.
        const int uvc = hough3_uv_cap(N, bx1 - bx0 + 1, by1 - by0 + 1);
        const bool fits = N <= HOUGH_BIG_LIST && 4 * (HOUGH3_FIXED_DWORDS + ((N + 3) & ~3) + 2 * uvc + HOUGH3_THREADS + N + 64) <= P.lds_large;
        return (fits && !P.acc_mode && !P.no_lds_acc && P.perm_big != nullptr) ? 2 : 3;
    }
    if (P.no_lds_acc) return 2;
    const int uvc = hough3_uv_cap(N, bx1 - bx0 + 1, by1 - by0 + 1);
    const int need = hough3_lds_need(N, uvc, cells, mask_dw);
    if (P.lds_tiny > 0 && need <= P.lds_tiny) return 4;
#if HOUGH_BAND_ROUTING
    const int rows = by1 - by0 + 1, nhb = (rows + HOUGH_HB_ROWS - 1) / HOUGH_HB_ROWS;
    // The launch the band windows might bring the frame into: the next smaller one than its corner windows ask for -- and only if
    // those miss it by less than half (the band windows save a third at best).  The scratch hough3_frame keeps the boxes in -- the
    // cells region of THAT launch -- must hold them: the same test as there, with that launch's budget.
    const int limit = (P.lds_tiny > 0 && need <= P.lds_tiny / 2 * 3) ? P.lds_tiny : P.lds_small;
    const int budget_dw = limit / 4 - HOUGH3_FIXED_DWORDS - ((N + 3) & ~3) - 2 * uvc - HOUGH3_THREADS;
    if (P.band_routing && need > limit && need <= limit / 2 * 3 && nhb <= HOUGH_CLS_MAXBANDS && scratch != nullptr && !P.acc_mode &&
        2 * nhb + HOUGH3_THREADS <= budget_dw - mask_dw) {
        unsigned* const bmin = scratch;
        unsigned* const bmax = scratch + HOUGH_CLS_MAXBANDS;
        for (int b = lane; b < nhb; b += 64) { bmin[b] = 0xFFFFu; bmax[b] = 0u; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int wlo = bx0 >> 6, wpr = (bx1 >> 6) - wlo + 1, TW = rows * wpr;
        const u64* dst = P.dst_bits + (size_t)i * ((size_t)P.H * P.WW);
#pragma unroll 2
        for (int e = lane; e < TW; e += 64) {
            const int ry = e / wpr, w = wlo + (e - ry * wpr);
            const u64 word = dst[(size_t)(by0 + ry) * P.WW + w];
            if (word) {
                atomicMin(&bmin[ry / HOUGH_HB_ROWS], (unsigned)(w * 64 + __builtin_ctzll(word)));
                atomicMax(&bmax[ry / HOUGH_HB_ROWS], (unsigned)(w * 64 + 63 - __builtin_clzll(word)));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // (kept small on purpose: this kernel's waves must find room beside the window kernel's -- no unrolling of the band loop)
        float c3[3], s3[3];
        int rmin3[3], rmax3[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int n = min(lane + 64 * q, NUM_ANGLE - 1);
            c3[q] = P.trig[2 * n];
            s3[q] = P.trig[2 * n + 1];
            rmin3[q] = 0x7FFFFFFF;
            rmax3[q] = -0x7FFFFFFF;
        }
#pragma unroll 1
        for (int b = 0; b < nhb; ++b) {
            const int x_lo = (int)bmin[b], x_hi = (int)bmax[b];
            if (x_lo > x_hi) continue;
            const int y_lo = by0 + b * HOUGH_HB_ROWS, y_hi = min(y_lo + HOUGH_HB_ROWS - 1, by1);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const bool cpos = c3[q] >= 0.f, spos = s3[q] >= 0.f;
                rmin3[q] = min(rmin3[q], rho_bin(cpos ? x_lo : x_hi, spos ? y_lo : y_hi, c3[q], s3[q], 0));
                rmax3[q] = max(rmax3[q], rho_bin(cpos ? x_hi : x_lo, spos ? y_hi : y_lo, c3[q], s3[q], 0));
            }
        }
        int cells_b = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int size = lane + 64 * q < NUM_ANGLE ? rmax3[q] - rmin3[q] + 1 : 0;
            cells_b += max(size, __shfl_xor(size, 1));
        }
        for (int o = 32; o > 0; o >>= 1) cells_b += __shfl_xor(cells_b, o);
        __builtin_amdgcn_wave_barrier();  // (the scratch is this wave's for the next frame too)
        if (hough3_lds_need(N, uvc, cells_b, mask_dw) <= limit) return limit == P.lds_small ? 0 : 4;
    }
#endif
    return need <= P.lds_small ? 0 : (need <= P.lds_large ? 1 : 2);
}

__global__ void __launch_bounds__(256) k_hough_classify(HoughParams P) {
    RAISE_WAVE_PRIORITY();
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= P.K) return;
    __shared__ unsigned cls_scratch[4][2 * HOUGH_CLS_MAXBANDS];
    const int tier = hough_classify_frame(P, i, lane, cls_scratch[threadIdx.x >> 6]);
    if (tier >= 0 && lane == 0) P.tier_list[tier * P.K + atomicAdd(&P.tier_count[tier], 1)] = i;
}

// ================================================================================================
// Three-wave PPHT: the same exact replay, one 192-thread workgroup per frame, thread t = angle t (180 of 192 lanes), so a
// vote / un-vote is ONE rho + ONE LDS atomic per lane and three SIMDs of the CU work on the same point.
//
// What makes it fast (the four-wave kernel of round 1 spent ~10 k cycles per group of 7.5 points):
//  * the VISITING ORDER is data-independent -- cv::RNG is seeded with the same constant for every call, and the swap-remove
//    of the point list never looks at pixel data -- so for an N-point frame the k-th visited point is list[perm_N[k]] with a
//    table computed once per process (k_perm_build).  No draws, no conflict analysis, no list writes in the kernel.
//  * 64 points are examined per step: one load of the table, one of the list, one look at the working mask; dead points
//    (cleared by an earlier segment: 3 of 4 in a streak) cost nothing more.
//  * the live points of a step vote in sub-groups of 16 back to back; a cell belongs to exactly one thread, whose LDS
//    atomics complete in program order, so every point sees exactly the sequential counts.  One barrier per sub-group (did any
//    angle of any wave reach the threshold); on a hit the later votes are taken back, the line is processed and the rest of
//    the step is re-examined against the updated mask -- identical to the one-point loop.
//  * the working mask lives in LDS (bit plane over the bounding box of the lit pixels; outside of it nothing is lit).
//  * a line: its two directions are walked by two waves at the same time; the cleared pixels are collected and ALL threads
//    take their votes back in one sweep (un-votes commute).
// Everything that decides control flow is computed identically by every wave from the same data.
struct Hough3 {
    const HoughParams& P;
    float tc, ts;          // this thread's angle
    unsigned vinc;         // 1 for a thread with an angle, 0 for the 12 spare lanes (their votes add nothing to a private cell)
    // Cells are 16 bits wide, two per dword, each biased by CELL_BIAS (a +-1 on one half never carries into the other).  Round 5:
    // the two halves of a dword belong to two ANGLES -- thread 2a owns the low halves, thread 2a + 1 the high halves of a window of
    // dwords the pair shares (adjacent angles: nearly the same rho range) -- so which half a vote goes to, the shifted increment and
    // the shift that extracts a returned count are per-thread constants instead of five vector instructions per vote (up to
    // round 4 a thread's window was a run of consecutive halves, and the parity of r + cbase picked the half).
    int cbase;             // LDS DWORD of rho bin r (before the +half shift): r + cbase
    int sh;                // 0 / 16: this thread's half
    unsigned vone;         // vinc << sh: one vote
    bool in_lds;           // HYBRID: this angle's window lives in LDS
    bool dead;             // HYBRID: the window does not, and no cell of it can ever hold `threshold` votes: the angle is out of the replay
    int* acc_row;          // HYBRID: this angle's row of the pooled global accumulator, shifted by half
    unsigned* cells;
    // working mask
    bool mask_lds;
    unsigned* lmask;       // [rows][mw] dwords over the bounding box
    int by0, rows, cx0, mw;
    u64* gmask;            // pooled global mask (whole frame) when the bounding box is too large
    int WW;
    int line_gap, frame, found, lane, wave, tid;
    unsigned* fix;         // fixed LDS block: [0..7] hit mailbox (2 parities x 4), [8..11] keys, [12..23] line ends, [24] un-vote count, [32..] scan
    unsigned* uv;
    bool touched_global;
    long long ph[8];       // steps (table, list, mask), votes, hits + barrier, trigger (take back, arg-max), walk, clear, un-vote, lines
    // full-accumulator mode (h3_accumulate): this thread's cached row maximum and whether an un-vote has hit its cell since
    int best_r;
    bool dirty;
};
// (TM: template parameter of the enclosing function -- the timed instantiations run only under MDH_HOUGH_DEBUG; the product's carry
// no clock reads, no phase counters and no scalar registers for them)
#define H3_T(C, k, t0) do { if (TM) { const long long t1_ = clock64(); (C).ph[k] += t1_ - (t0); (t0) = t1_; } } while (0)

template <bool HYBRID>
__device__ __forceinline__ bool h3_bit(const Hough3& C, int x, int y) {
    if (!HYBRID || C.mask_lds) {
        const int ry = y - C.by0, cx = (x >> 5) - C.cx0;
        const bool inb = (unsigned)ry < (unsigned)C.rows && (unsigned)cx < (unsigned)C.mw;
        const unsigned v = C.lmask[inb ? ry * C.mw + cx : 0];
        return inb && ((v >> (x & 31)) & 1);
    }
    return (ld_l2(&C.gmask[(size_t)y * C.WW + (x >> 6)]) >> (x & 63)) & 1;
}
// clear the bit, return whether it was set
template <bool HYBRID>
__device__ __forceinline__ bool h3_clear(const Hough3& C, int x, int y) {
    const unsigned bit = 1u << (x & 31);
    if (!HYBRID || C.mask_lds) {
        const int ry = y - C.by0, cx = (x >> 5) - C.cx0;
        const bool inb = (unsigned)ry < (unsigned)C.rows && (unsigned)cx < (unsigned)C.mw;
        if (!inb) return false;
        return atomicAnd(&C.lmask[ry * C.mw + cx], ~bit) & bit;
    }
    return atomicAnd(&((unsigned*)C.gmask)[((size_t)y * C.WW + (x >> 6)) * 2 + ((x >> 5) & 1)], ~bit) & bit;
}

// `inc` = 1 for a real vote, 0 for an empty slot of a sub-group (the votes of a sub-group are issued branch-free so that the
// sixteen independent rho chains interleave; an empty slot adds nothing to the cell of point (0, 0))
// The vote is split in two so that the atomics of a sub-group are all in flight before the first returned count is looked at.
template <bool HYBRID>
__device__ __forceinline__ unsigned h3_vote_issue(Hough3& C, float x, float y, unsigned inc, int& sh) {
    // (x, y: the pixel's coordinates ALREADY as floats -- exact, they are below 2^15: converted once per point by the lane that
    // holds it instead of once per vote by every one of the 192 threads, round 3)
    const int r = __float2int_rn(__fadd_rn(__fmul_rn(x, C.tc), __fmul_rn(y, C.ts)));
    if (!HYBRID || C.in_lds) {
        sh = C.sh;
        return atomicAdd(&C.cells[r + C.cbase], C.vone & (0u - inc));
    }
    sh = -1;
    if (C.dead) return (unsigned)(-KEY_BIAS - 2);  // (read as a count: below every threshold)
    return (unsigned)atomicAdd(&C.acc_row[r], (int)inc);
}
template <bool HYBRID>
__device__ __forceinline__ int h3_vote_value(unsigned o, int sh) {
    if (HYBRID && sh < 0) return (int)o;
    return (int)((o >> sh) & 0xFFFFu) - CELL_BIAS;
}
__device__ __forceinline__ u64 h3_uniform(u64 v) {  // the value is wave-uniform: tell the compiler (keeps it in SGPRs)
    return ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}
template <bool HYBRID, bool ACC = false>
__device__ __forceinline__ void h3_unvote(Hough3& C, float x, float y) {
    const int r = __float2int_rn(__fadd_rn(__fmul_rn(x, C.tc), __fmul_rn(y, C.ts)));
    if (ACC && r == C.best_r) C.dirty = true;
    if (!HYBRID || C.in_lds) {
        atomicSub(&C.cells[r + C.cbase], C.vone);
    } else if (!C.dead) {
        C.touched_global = true;
        atomicSub(&C.acc_row[r], 1);
    }
}

// exclusive scan over the workgroup (3 waves); `total` = sum over all threads.  Two barriers.
__device__ __forceinline__ int h3_scan(int v, int& total, unsigned* scratch, int lane, int wave) {
    // inclusive wave scan on DPP moves (Hillis-Steele inside a 16-lane row, then the row totals carried on with row_bcast:15 / :31)
    // instead of six ds_bpermute round trips (__shfl_up)
    int incl = v;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);  // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);  // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);  // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);  // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    if (lane == 63) scratch[wave] = (unsigned)incl;
    __syncthreads();
    const int s0 = (int)scratch[0], s1 = (int)scratch[1], s2 = (int)scratch[2];
    total = s0 + s1 + s2;
    const int base = wave == 0 ? 0 : (wave == 1 ? s0 : s0 + s1);
    __syncthreads();
    return base + incl - v;
}

// The point (x, y) reached the threshold at angle bn (hough.cpp, second half of the loop).  Wave k (k = 0, 1) walks direction
// k; ends with a barrier after which every wave sees the cleared mask and the taken-back votes of ITS OWN cells.
// ACC (full-accumulator mode): every lit pixel has voted, so the votes of every cleared pixel are taken back whether or not
// the segment is long enough -- the accumulator stays the vote count of the pixels still in the mask.
template <bool HYBRID, bool TM, bool ACC = false>
__device__ void h3_line(Hough3& C, int x, int y, int bn) {
    const HoughParams& P = C.P;
    const int lane = C.lane;
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
    const int k = C.wave;  // direction of this wave (wave 2 has none)
    const int dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
    unsigned* ends = C.fix + 12;  // [k][end_step, end_x, end_y]
    long long tph = TM ? clock64() : 0;
    if (TM) C.ph[7] += 1;  // [15:0] line events, [39:16] vote slots issued, [63:40] sub-groups
    if (C.tid == 0) C.fix[24] = 0;  // un-vote list is empty (its last readers passed a barrier since)
    if (k < 2) {
        int gap = 0, last_lit = 0;
        bool done = false;
        for (int s0 = 0; !done; s0 += 64) {
            const int s = s0 + lane;
            const int X = x0 + dx * s, Y = y0 + dy * s;
            const int j1 = xflag ? X : X >> 16, i1 = xflag ? Y >> 16 : Y;
            const bool inside = j1 >= 0 && j1 < P.W && i1 >= 0 && i1 < P.H;
            bool lit = false;
            if (inside) lit = h3_bit<HYBRID>(C, j1, i1);
            const u64 mi = __ballot(inside), ml = __ballot(lit);
            // scan the 64 steps run by run (lit runs reset the gap, dark runs extend it) instead of bit by bit
            const int n_in = mi == ~0ull ? 64 : __builtin_ctzll(~mi);  // steps before the image border
            int pos = 0;
            while (pos < n_in) {
                const u64 rest = ml >> pos;
                if (rest & 1) {
                    const u64 inv = ~rest;
                    int run = inv ? __builtin_ctzll(inv) : 64;
                    run = min(run, n_in - pos);
                    pos += run;
                    last_lit = s0 + pos - 1;
                    gap = 0;
                } else {
                    int run = rest ? __builtin_ctzll(rest) : 64;
                    run = min(run, n_in - pos);
                    if (gap + run > C.line_gap) { done = true; break; }
                    gap += run;
                    pos += run;
                }
            }
            if (!done && n_in < 64) done = true;  // left the image
        }
        if (lane == 0) {
            const int X = x0 + dx * last_lit, Y = y0 + dy * last_lit;
            ends[3 * k + 0] = (unsigned)last_lit;
            ends[3 * k + 1] = (unsigned)(xflag ? X : X >> 16);
            ends[3 * k + 2] = (unsigned)(xflag ? Y >> 16 : Y);
        }
    }
    __syncthreads();
    H3_T(C, 4, tph);
    const int ex0 = __builtin_amdgcn_readfirstlane((int)ends[1]), ey0 = __builtin_amdgcn_readfirstlane((int)ends[2]);
    const int ex1 = __builtin_amdgcn_readfirstlane((int)ends[4]), ey1 = __builtin_amdgcn_readfirstlane((int)ends[5]);
    const bool good = abs(ex1 - ex0) >= P.line_length || abs(ey1 - ey0) >= P.line_length;
    if (k < 2) {
        const int end_step = __builtin_amdgcn_readfirstlane((int)ends[3 * k]);
        for (int s0 = 0; s0 <= end_step; s0 += 64) {
            const int s = s0 + lane;
            bool was = false;
            unsigned pt = 0;
            if (s <= end_step) {
                const int X = x0 + dx * s, Y = y0 + dy * s;
                const int j1 = xflag ? X : X >> 16, i1 = xflag ? Y >> 16 : Y;
                was = h3_clear<HYBRID>(C, j1, i1);  // (step 0 belongs to both directions: the atomic gives it to one of them)
                pt = ((unsigned)i1 << 16) | (unsigned)j1;
            }
            if (good || ACC) {
                const u64 ms = __ballot(was);
                if (ms) {
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(&C.fix[24], (unsigned)__popcll(ms));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (was) {  // (x, y) as floats: the sweep below is run by every thread for every entry
                        const unsigned slot2 = 2 * (base + __popcll(ms & ((1ull << lane) - 1)));
                        C.uv[slot2] = __float_as_uint((float)(pt & 0xFFFF));
                        C.uv[slot2 + 1] = __float_as_uint((float)(pt >> 16));
                    }
                }
            }
        }
        if (HYBRID && !C.mask_lds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();  // the mask is cleared, the un-vote list is complete
    H3_T(C, 5, tph);
    if (good || ACC) {
        const int M = __builtin_amdgcn_readfirstlane((int)C.fix[24]);
        int m = 0;
        for (; m + 2 <= M; m += 2) {  // two points per LDS read (the list is 16-byte aligned)
            const uint4 q = *(const uint4*)&C.uv[2 * m];
            h3_unvote<HYBRID, ACC>(C, __uint_as_float(q.x), __uint_as_float(q.y));
            h3_unvote<HYBRID, ACC>(C, __uint_as_float(q.z), __uint_as_float(q.w));
        }
        for (; m < M; ++m) h3_unvote<HYBRID, ACC>(C, __uint_as_float(C.uv[2 * m]), __uint_as_float(C.uv[2 * m + 1]));
    }
    if (good) {
        if (C.found < MAX_LINES && C.tid == 0) {
            int* o = P.out_lines + ((size_t)C.frame * MAX_LINES + C.found) * 4;
            o[0] = ex0; o[1] = ey0; o[2] = ex1; o[3] = ey1;
        }
        ++C.found;
    }
    H3_T(C, 6, tph);
}

// The count a thread's cell holds right now for point (x, y) (its own earlier operations on the cell have completed: LDS
// operations of a wave execute in order; a returning atomic on the thread's own row of the pooled accumulator is ordered
// behind its earlier atomics on the same address).
template <bool HYBRID>
__device__ __forceinline__ int h3_cell_now(Hough3& C, float x, float y) {
    const int r = __float2int_rn(__fadd_rn(__fmul_rn(x, C.tc), __fmul_rn(y, C.ts)));
    if (!HYBRID || C.in_lds) return (int)((C.cells[r + C.cbase] >> C.sh) & 0xFFFFu) - CELL_BIAS;
    if (C.dead) return -KEY_BIAS;  // below the winner's count (>= threshold), whatever the angle's true count (< threshold) is
    return atomicAdd(&C.acc_row[r], 0);
}

#ifndef H3_SUB
#define H3_SUB 8  // largest sub-group (A/B builds: -DH3_SUB=16 is round 2's)
#endif
template <bool HYBRID, bool TM>
__device__ void h3_replay(Hough3& C, int N, const unsigned* list, const unsigned short* perm) {
    const HoughParams& P = C.P;
    const int lane = C.lane;
    unsigned round = 0;
    int gsz = H3_SUB;
    unsigned nxt = lane < N ? perm[lane] : 0;  // the table entries of the following step are fetched one step ahead
    for (int k0 = 0; k0 < N; k0 += 64) {
        long long tph = TM ? clock64() : 0;
        const int kk = k0 + lane;
        const unsigned idx = nxt;
        nxt = kk + 64 < N ? perm[kk + 64] : 0;
        const unsigned pt = kk < N ? list[idx] : 0;
        const int px = pt & 0xFFFF, py = pt >> 16;
        const int fxb = (int)__float_as_uint((float)px), fyb = (int)__float_as_uint((float)py);  // the same as float bit patterns
        u64 am = __ballot(kk < N && h3_bit<HYBRID>(C, px, py));
        H3_T(C, 0, tph);
        while (am) {
            // sub-group: the first (up to) `gsz` live points of the step vote back to back.  Votes behind a trigger are wasted
            // (taken back and repeated), and triggers come in bursts (the parallel lines of one streak), so the sub-group
            // shrinks to 4 after a trigger and doubles with every quiet sub-group.
            // Registers (round 3): the kernel shares its SIMDs with the window kernel's waves (148 VGPRs, three per SIMD: 56
            // registers are left), so a sub-group keeps the raw returns of its atomics only until they are turned into one hit
            // bit each (round 2 kept sixteen returned counts + sixteen shifts alive across the barrier: 87 VGPRs, and every SIMD
            // that hosted a Hough wave lost a window wave).  The one count that is needed later -- the trigger point's, for the
            // arg-max over the angles -- is read back from the cell after the votes behind the trigger have been taken back.
            am = h3_uniform(am);
            u64 rem = am;
            const int gused = gsz <= 4 ? 4 : (gsz <= 8 || H3_SUB <= 8 ? 8 : 16);  // slots of this sub-group
            if (TM) C.ph[7] += ((long long)min(gused, (int)__popcll(am)) << 16) + (1ll << 40);
            unsigned hits = 0;
            {
                unsigned raw[H3_SUB];
                unsigned hasm = 0;
#pragma unroll
                for (int j = 0; j < H3_SUB; ++j) {
                    if (j < gused) {  // (gused is wave-uniform: 4, 8 or 16 slots)
                        const bool has = rem != 0;
                        const int l = has ? __builtin_ctzll(rem) : 0;
                        rem &= rem - 1;
                        hasm |= (has ? 1u : 0u) << j;
                        int sh;
                        raw[j] = h3_vote_issue<HYBRID>(C, __uint_as_float((unsigned)__builtin_amdgcn_readlane(fxb, l)),
                                                      __uint_as_float((unsigned)__builtin_amdgcn_readlane(fyb, l)), has ? 1u : 0u, sh);
                    }
                }
                if (TM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                H3_T(C, 1, tph);
                // most sub-groups are quiet: ONE test on the largest returned count says so; the per-slot tests only run on a hit
                int val[H3_SUB];
                int vmax = -KEY_BIAS;
#pragma unroll
                for (int j = 0; j < H3_SUB; ++j) {
                    val[j] = -KEY_BIAS;
                    if (j < gused && ((hasm >> j) & 1)) {
                        if (!HYBRID || C.in_lds) val[j] = (int)((raw[j] >> C.sh) & 0xFFFFu) - CELL_BIAS;
                        else val[j] = (int)raw[j];
                        vmax = max(vmax, val[j]);
                    }
                }
                if (__any(vmax + 1 >= P.threshold)) {
#pragma unroll
                    for (int j = 0; j < H3_SUB; ++j)
                        if (__any(val[j] + 1 >= P.threshold)) hits |= 1u << j;
                }
            }
            if (lane == 0) C.fix[(round & 1) * 4 + C.wave] = hits;
            __syncthreads();
            const unsigned Hm = (unsigned)__builtin_amdgcn_readfirstlane(
                (int)(C.fix[(round & 1) * 4] | C.fix[(round & 1) * 4 + 1] | C.fix[(round & 1) * 4 + 2]));
            ++round;
            H3_T(C, 2, tph);
            if (!Hm) {
                am = rem;
                gsz = min(H3_SUB, gsz * 2);
                continue;
            }
            gsz = 4;
            const int trig = __builtin_ctz(Hm);
            // lane of the trigger point; the sub-group's points after it voted too early: take those votes back
            u64 walk = am;
            int tl = 0;
#pragma unroll
            for (int j = 0; j < H3_SUB; ++j) {
                if (walk) {
                    const int l = __builtin_ctzll(walk);
                    walk &= walk - 1;
                    if (j == trig) tl = l;
                    else if (j > trig && j < gused) h3_unvote<HYBRID>(C, __uint_as_float((unsigned)__builtin_amdgcn_readlane(fxb, l)),
                                                                        __uint_as_float((unsigned)__builtin_amdgcn_readlane(fyb, l)));
                }
            }
            const int tx = __builtin_amdgcn_readlane(px, tl), ty = __builtin_amdgcn_readlane(py, tl);
            // the count the trigger point's vote produced in this thread's cell: earlier points' votes + its own, now that the
            // later ones are gone (= the `val` of hough.cpp's scan over n)
            const int kt1 = h3_cell_now<HYBRID>(C, (float)tx, (float)ty);
            const unsigned key = ((unsigned)(kt1 + KEY_BIAS) << 8) | (unsigned)(255 - (C.tid < 256 ? C.tid : 255));
            const unsigned wmax = wave_max_u32(C.vinc ? key : 0u);
            if (lane == 0) C.fix[8 + C.wave] = wmax;
            __syncthreads();
            const unsigned best = (unsigned)__builtin_amdgcn_readfirstlane((int)max(max(C.fix[8], C.fix[9]), C.fix[10]));
            H3_T(C, 3, tph);
            h3_line<HYBRID, TM>(C, tx, ty, 255 - (int)(best & 0xFF));  // ends with a barrier
            if (TM) tph = clock64();
            // the rest of the step against the updated mask
            am = __ballot(kk < N && lane > tl && h3_bit<HYBRID>(C, px, py));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// hough_mode = MDH_HOUGH_ACCUMULATOR: a FULL-accumulator Hough on the same data structures (BASELINE north_star's
// "LDS-binned atomics").  Every lit pixel votes every angle at once (order-free LDS atomics, thread t = angle t), then the
// strongest cell of the whole accumulator is taken again and again while it holds at least `threshold` votes: a lit pixel on
// that (rho, theta) line seeds the same walk / clear / take-back procedure as the progressive transform (h3_line).  No random
// visiting order, no dependence on which cell happens to cross the threshold first: the result is a function of the binary
// image alone, the strongest lines come out first, and the work per frame is one vote sweep plus one line procedure per
// segment.  It is a DIFFERENT algorithm from cv2.HoughLinesP: its segments are compared with the exact replay's under SURVEY
// 8d's tolerance rule (rho / theta of matched segments, collector records), never bit for bit.
__device__ __forceinline__ int h3_cell_value(const Hough3& C, int r) {
    return (int)((C.cells[r + C.cbase] >> C.sh) & 0xFFFFu) - CELL_BIAS;
}
template <bool HYBRID, bool TM>
__device__ void h3_accumulate(Hough3& C, int N, const unsigned* list, int w_lo, int w_size) {
    const HoughParams& P = C.P;
    const int lane = C.lane, tid = C.tid;
    // ---- one sweep: all votes (un-ordered; a thread's own cells, but a dword can hold the last cell of one angle's window and
    // the first of the next, hence atomics)
    {
        int p = 0;
        for (; p + 4 <= N; p += 4) {
            const uint4 q = *(const uint4*)&list[p];
            int sh;
            h3_vote_issue<HYBRID>(C, (float)(q.x & 0xFFFF), (float)(q.x >> 16), 1u, sh);
            h3_vote_issue<HYBRID>(C, (float)(q.y & 0xFFFF), (float)(q.y >> 16), 1u, sh);
            h3_vote_issue<HYBRID>(C, (float)(q.z & 0xFFFF), (float)(q.z >> 16), 1u, sh);
            h3_vote_issue<HYBRID>(C, (float)(q.w & 0xFFFF), (float)(q.w >> 16), 1u, sh);
        }
        for (; p < N; ++p) {
            const unsigned q = list[p];
            int sh;
            h3_vote_issue<HYBRID>(C, (float)(q & 0xFFFF), (float)(q >> 16), 1u, sh);
        }
    }
    if (HYBRID) __threadfence();
    __syncthreads();
    C.dirty = true;
    C.best_r = 0;
    int best = -1;
    const bool has_angle = tid < NUM_ANGLE;
    for (int iter = 0; iter <= N; ++iter) {  // every line event clears at least its seed pixel
        // ---- this angle's strongest cell (smallest rho among equals); rescanned only when an un-vote has touched it
        if (has_angle && C.dirty) {
            best = -1;
            int br = w_lo;
            for (int r = w_lo; r < w_lo + w_size; ++r) {
                int v;
                if (!HYBRID || C.in_lds) v = h3_cell_value(C, r);
                else v = (int)ld_l2((const unsigned*)&C.acc_row[r]);
                if (v > best) {
                    best = v;
                    br = r;
                }
            }
            C.best_r = br;
            C.dirty = false;
        }
        // ---- strongest cell of the accumulator: larger count first, then the smaller angle ("first maximum wins")
        const unsigned key = has_angle ? (((unsigned)(best + 1 + KEY_BIAS) << 8) | (unsigned)(255 - tid)) : 0u;
        const unsigned wmax = wave_max_u32(key);
        if (lane == 0) C.fix[8 + C.wave] = wmax;
        __syncthreads();
        const unsigned top = (unsigned)__builtin_amdgcn_readfirstlane((int)max(max(C.fix[8], C.fix[9]), C.fix[10]));
        const int count = (int)(top >> 8) - 1 - KEY_BIAS;
        const int bn = 255 - (int)(top & 0xFF);
        if (count < P.threshold) break;  // (uniform: every thread reads the same three words)
        if (tid == bn) C.fix[11] = (unsigned)C.best_r;
        if (tid == 0) C.fix[25] = 0xFFFFFFFFu;
        __syncthreads();
        const int rho = (int)C.fix[11];
        // ---- seed: the MIDDLE one (row-major order) of the lit pixels that vote for that cell.  The walk follows the cell's angle
        // (1-degree grid) from the seed in both directions; starting in the middle halves the drift against a line whose true
        // angle lies between two grid angles.
        const float sc = P.trig[2 * bn], ss = P.trig[2 * bn + 1];
        const int chunk = (N + HOUGH3_THREADS - 1) / HOUGH3_THREADS;
        const int p_lo = min(N, tid * chunk), p_hi = min(N, p_lo + chunk);
        auto votes_here = [&](int p) -> bool {
            const unsigned q = list[p];
            const int x = q & 0xFFFF, y = q >> 16;
            return __float2int_rn(__fadd_rn(__fmul_rn((float)x, sc), __fmul_rn((float)y, ss))) == rho && h3_bit<HYBRID>(C, x, y);
        };
        int mine = 0;
        for (int p = p_lo; p < p_hi; ++p) mine += votes_here(p) ? 1 : 0;
        int voters;
        const int before = h3_scan(mine, voters, C.fix + 32, lane, C.wave);  // (two barriers)
        const int want = voters / 2;
        if (voters > 0 && want >= before && want < before + mine) {
            int k = want - before;
            for (int p = p_lo; p < p_hi; ++p)
                if (votes_here(p) && k-- == 0) {
                    C.fix[25] = (unsigned)p;
                    break;
                }
        }
        __syncthreads();
        const unsigned seed = C.fix[25];
        __syncthreads();
        if (seed == 0xFFFFFFFFu) {
            // cannot happen while the accumulator mirrors the mask; kept so that a stale cell could never spin the loop
            if (tid == bn) {
                C.dirty = true;
                if (!HYBRID || C.in_lds) {
                    atomicAnd(&C.cells[rho + C.cbase], ~(0xFFFFu << C.sh));
                    atomicOr(&C.cells[rho + C.cbase], (unsigned)CELL_BIAS << C.sh);
                } else {
                    C.acc_row[rho] = 0;
                }
            }
            __syncthreads();
            continue;
        }
        const unsigned q = list[seed];
        h3_line<HYBRID, TM, true>(C, q & 0xFFFF, q >> 16, bn);  // walks, clears, takes the votes back (marks touched maxima dirty)
        if (HYBRID) __threadfence();
        __syncthreads();
    }
}

template <bool HYBRID, bool TM>
__device__ void hough3_frame(const HoughParams& P, int i, int slot, unsigned* lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = P.litcount[i];
    const long long w0 = TM && P.dbg ? wall_clock64() : 0;
    unsigned* fix = lds;
    unsigned* list = lds + HOUGH3_FIXED_DWORDS;
    const int Np = (N + 3) & ~3;  // keeps the un-vote list and the mask 16-byte aligned
    unsigned* uv = list + Np;
    const int bx0 = P.W - 1 - P.bbox[4 * i], bx1 = P.bbox[4 * i + 1], by0 = P.H - 1 - P.bbox[4 * i + 2], by1 = P.bbox[4 * i + 3];
    const int uvc = hough3_uv_cap(N, bx1 - bx0 + 1, by1 - by0 + 1);
    unsigned* lmask = uv + 2 * uvc;
    const int wlo = bx0 >> 6, wpr = (bx1 >> 6) - wlo + 1, rows = by1 - by0 + 1;
    const int mask_dw = rows * wpr * 2;
    const int budget_dw = P.lds_bytes / 4 - HOUGH3_FIXED_DWORDS - Np - 2 * uvc - HOUGH3_THREADS;  // mask + cells
    bool mask_lds = true;
    const bool big_n = HYBRID && N > HOUGH_MAX_LIST;  // (its own visiting-order table, built below in the region behind the un-vote list)
    if (HYBRID) mask_lds = !big_n && mask_dw <= budget_dw / 4;  // a huge bounding box: the mask stays in global memory, the LDS goes to the cells
    unsigned* cells = lmask + (mask_lds ? mask_dw : 0);
    // halves the cells may take (the all-LDS launches: the routing made sure the frame's windows fit -- its corner windows, or, round 6,
    // its band windows: hough_classify_frame)
    const int cell_cap = HYBRID ? (P.no_lds_acc ? 0 : 2 * (budget_dw - (mask_lds ? mask_dw : 0))) : 2 * (budget_dw - mask_dw);
    if (tid < HOUGH3_FIXED_DWORDS) fix[tid] = 0;
    // ---- row-major point list + working mask from the words of the bounding box
    const size_t words = (size_t)P.H * P.WW;
    const u64* dst = P.dst_bits + (size_t)i * words;
    u64* gmask = P.work_bits + (size_t)slot * words;
    const int TW = rows * wpr;
    __syncthreads();
    if (mask_lds) {
        // the words of the bounding box into the LDS mask first, many loads in flight (round 3: the loop below used to fetch its
        // word itself -- one dependent HBM round trip per 192 words between two barriers: 50 of a 2000-point frame's 60 us of set-up)
#pragma unroll 4
        for (int e = tid; e < TW; e += HOUGH3_THREADS) {
            const int ry = e / wpr;
            const u64 word = dst[(size_t)(by0 + ry) * P.WW + wlo + (e - ry * wpr)];
            lmask[2 * e] = (unsigned)word;
            lmask[2 * e + 1] = (unsigned)(word >> 32);
        }
        __syncthreads();
    }
    int run = 0;
    for (int base = 0; base < TW; base += HOUGH3_THREADS) {
        const int e = base + tid;
        u64 word = 0;
        int y = 0, w = 0;
        if (e < TW) {
            const int ry = e / wpr;
            y = by0 + ry;
            w = wlo + (e - ry * wpr);
            if (mask_lds) {
                word = (u64)lmask[2 * e] | ((u64)lmask[2 * e + 1] << 32);
            } else {
                word = dst[(size_t)y * P.WW + w];
                if (word) gmask[(size_t)y * P.WW + w] = word;
            }
        }
        int tot;
        int pos = run + h3_scan(__popcll(word), tot, fix + 32, lane, wave);
        while (word) {
            const int bpos = __builtin_ctzll(word);
            word &= word - 1;
            list[pos++] = ((unsigned)y << 16) | (unsigned)(w * 64 + bpos);
        }
        run += tot;
    }
    if (HYBRID && !mask_lds) {
        __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const unsigned short* perm = P.perm + (size_t)N * (N - 1) / 2;
    if (big_n) {
        // The visiting order of cv::HoughLinesP for THIS list length (k_perm_build's job for N <= HOUGH_MAX_LIST, once per process):
        // idx_k = rnd_k % (N - k) for all k at once, then the swap-remove chain by one lane on 16-bit LDS arrays (N steps of three
        // dependent LDS accesses: ~0.15 us each), written to the workgroup's slot of perm_big.  The scratch -- 2 N half-words --
        // is the region the mask and the cells would take; the mask of such a frame stays in global memory.
        unsigned short* const a = (unsigned short*)lmask;
        unsigned short* const ix = a + ((N + 1) & ~1);
        unsigned short* const out = P.perm_big + (size_t)slot * HOUGH_BIG_LIST;
        for (int k = tid; k < N; k += HOUGH3_THREADS) {
            a[k] = (unsigned short)k;
            ix[k] = (unsigned short)(P.rnd[k] % (unsigned)(N - k));
        }
        __syncthreads();
        if (tid == 0) {
            for (int k = 0; k < N; ++k) {
                const int idx = ix[k];
                const unsigned short v = a[idx];
                a[idx] = a[N - 1 - k];
                out[k] = v;
            }
        }
        __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        __threadfence();  // (every wave: the table was written through the L2 by one lane; this CU's vector cache may hold the slot's previous table)
        perm = out;
    }
    // ---- rho windows from the bounding box: rho is monotone in x and in y for a fixed angle, so its extremes are at the corners
    int w_lo = 0, w_size = 0;
    float tc = 0.f, ts = 0.f;
    if (tid < NUM_ANGLE) {
        tc = P.trig[2 * tid];
        ts = P.trig[2 * tid + 1];
        const int r00 = rho_bin(bx0, by0, tc, ts, 0), r10 = rho_bin(bx1, by0, tc, ts, 0);
        const int r01 = rho_bin(bx0, by1, tc, ts, 0), r11 = rho_bin(bx1, by1, tc, ts, 0);
        w_lo = min(min(r00, r10), min(r01, r11));
        w_size = max(max(r00, r10), max(r01, r11)) - w_lo + 1;
    }
    // storage: angles 2a and 2a + 1 share a window of dwords (low / high halves), as long as the larger of their two windows --
    // both threads of a pair ask for `w_st` halves, so every decision below (fits / goes to the global accumulator) is the pair's
    int w_st = max(w_size, __shfl_xor(w_size, 1));
    int placed = w_st, total = 0;
    auto band_windows = [&]() {
        {
            // The corner windows do not fit.  A streak is thin and usually oblique, so the rho range its points really span is
            // up to a third smaller than what the corners of its bounding box span.  Round 2 swept the point list for the exact
            // window of every angle (the cost of voting every point once: 40 of a 2100-point frame's 69 us of set-up); round 3
            // takes the x-extent of the points per band of HB_ROWS rows (one LDS min / max per point, all threads) and lets every
            // angle walk the bands' boxes instead of the points: rho is monotone in x and in y -- also in float32, rounding is
            // monotone -- so a box's extremes are at two of its corners.  The windows come out at most HB_ROWS - 1 rows' worth
            // of rho wider than the exact ones (a superset is all the replay needs); the cells region is free until it is
            // initialised below and serves as scratch.
            constexpr int HB_ROWS = HOUGH_HB_ROWS;
            const int nhb = (rows + HB_ROWS - 1) / HB_ROWS;
            unsigned* const bmin = cells;
            unsigned* const bmax = cells + nhb;
            if (2 * nhb + HOUGH3_THREADS <= cell_cap / 2) {
                for (int b = tid; b < nhb; b += HOUGH3_THREADS) { bmin[b] = 0xFFFFu; bmax[b] = 0u; }
                __syncthreads();
                for (int p = tid; p < N; p += HOUGH3_THREADS) {
                    const unsigned q = list[p];
                    const int b = ((int)(q >> 16) - by0) / HB_ROWS;
                    atomicMin(&bmin[b], q & 0xFFFFu);
                    atomicMax(&bmax[b], q & 0xFFFFu);
                }
                __syncthreads();
                if (tid < NUM_ANGLE) {
                    int rmin = 0x7FFFFFFF, rmax = -0x7FFFFFFF;
                    const bool cpos = tc >= 0.f, spos = ts >= 0.f;
                    // (the all-LDS kernel lives on 54 registers -- a wave of it beside three window waves: no unrolling there)
#pragma unroll(HYBRID ? 4 : 1)
                    for (int b = 0; b < nhb; ++b) {
                        const int x_lo = (int)bmin[b], x_hi = (int)bmax[b];
                        if (x_lo > x_hi) continue;  // (no point in this band: the same test for every thread)
                        const int y_lo = by0 + b * HB_ROWS, y_hi = min(y_lo + HB_ROWS - 1, by1);
                        rmin = min(rmin, rho_bin(cpos ? x_lo : x_hi, spos ? y_lo : y_hi, tc, ts, 0));
                        rmax = max(rmax, rho_bin(cpos ? x_hi : x_lo, spos ? y_hi : y_lo, tc, ts, 0));
                    }
                    w_lo = rmin;
                    w_size = rmax - rmin + 1;
                }
                __syncthreads();  // (the scratch is read for the last time above; the size search below reuses it)
            } else if (tid < NUM_ANGLE) {  // (no room for the boxes: the exact sweep)
                int rmin = 0x7FFFFFFF, rmax = -0x7FFFFFFF;
#pragma unroll(HYBRID ? 4 : 1)
                for (int p = 0; p < N; ++p) {
                    const unsigned q = list[p];
                    const int r = rho_bin(q & 0xFFFF, q >> 16, tc, ts, 0);
                    rmin = min(rmin, r);
                    rmax = max(rmax, r);
                }
                w_lo = rmin;
                w_size = rmax - rmin + 1;
            }
            w_st = max(w_size, __shfl_xor(w_size, 1));
            placed = w_st;
        }
    };
    if (HYBRID) {
        h3_scan(w_st, total, fix + 32, lane, wave);
        if (total > cell_cap && cell_cap > 0) {
            band_windows();
            h3_scan(w_st, total, fix + 32, lane, wave);
        }
        if (total > cell_cap) {
            // still too many: the smallest windows first (the angles across the streak, where the votes pile up) -- the largest
            // size limit that fits; the other angles use the pooled global accumulator.  Every thread adds up the windows not
            // larger than its own (180 LDS reads, one barrier) instead of a bisection with a workgroup-wide scan per step.
            unsigned* const wsz = cells;
            if (HOUGH3_THREADS + 1 <= cell_cap / 2) {
                wsz[tid] = (unsigned)w_st;
                if (tid == 0) wsz[HOUGH3_THREADS] = 0;
                __syncthreads();
                int mine = 0;
                for (int j = 0; j < NUM_ANGLE; ++j) {
                    const int wj = (int)wsz[j];
                    mine += wj <= w_st ? wj : 0;
                }
                if (tid < NUM_ANGLE && mine <= cell_cap) atomicMax(&wsz[HOUGH3_THREADS], (unsigned)w_st);
                __syncthreads();
                const int lo = (int)wsz[HOUGH3_THREADS];
                placed = w_st <= lo ? w_st : 0;
                __syncthreads();
            } else {
                int lo = 0, hi = P.numrho;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    int sum;
                    h3_scan(w_st <= mid ? w_st : 0, sum, fix + 32, lane, wave);
                    if (sum <= cell_cap) lo = mid; else hi = mid - 1;
                }
                placed = w_st <= lo ? w_st : 0;
            }
        }
    }
    int off = h3_scan(placed, total, fix + 32, lane, wave);  // (in halves; both threads of a pair placed the same amount, so a pair starts on a dword)
    if (!HYBRID && total > cell_cap) {
        // the routing sent this frame here on its band windows (round 6): the corner windows do not fit this launch's LDS
        band_windows();
        off = h3_scan(placed, total, fix + 32, lane, wave);
        if (total > cell_cap) __builtin_trap();  // (the routing and this kernel disagree: never go on into the neighbour's LDS)
    }
    const int cell_dw = (total + 1) / 2;
    // ---- initialise: cells at the bias, one private zero cell per thread for the spare lanes
    for (int p = tid; p < cell_dw; p += HOUGH3_THREADS) cells[p] = (CELL_BIAS << 16) | CELL_BIAS;
    cells[cell_dw + tid] = 0;
    __syncthreads();
    Hough3 C{P, tc, ts, tid < NUM_ANGLE ? 1u : 0u, 0, 16 * (tid & 1), (tid < NUM_ANGLE ? 1u : 0u) << (16 * (tid & 1)), true, false, nullptr, cells, mask_lds, lmask, by0, rows, 2 * wlo, 2 * wpr, gmask, P.WW,
             (int)rint(P.out_gap[i]), i, 0, lane, wave, tid, fix, uv, false, {0, 0, 0, 0, 0, 0, 0, 0}};
    if (tid < NUM_ANGLE) {
        C.in_lds = placed > 0;
        C.cbase = (((tid & 1) ? off - placed : off) >> 1) - w_lo;  // the pair's first dword, rho bin w_lo of THIS angle on it
        if (HYBRID) C.acc_row = P.accum + (size_t)slot * NUM_ANGLE * P.numrho + (size_t)tid * P.numrho + P.half;
    } else {
        C.cbase = cell_dw + tid;  // r is 0 for the spare lanes (tc = ts = 0): their private dword
    }
    if (HYBRID && !P.acc_mode && !P.no_lds_acc) {
        // Angles whose window went to the pooled global accumulator (the LARGEST windows: the directions across which the streak
        // spreads its points thinly).  Every vote of the replay would wait for their returning global atomics (round 2: 40 % of a
        // hybrid frame's time at the sub-group barriers).  But a cell never holds more votes than the number of points that map to
        // it -- votes come from points only, un-votes only lower a count -- so ONE order-free sweep (non-returning atomics, all
        // points) gives every cell's upper bound, and an angle whose largest cell stays below the threshold can never trigger,
        // never be the arg-max of a trigger (that one holds >= threshold), and is dropped from the replay altogether: the frame
        // then runs at LDS speed.  An angle that could trigger keeps the global path (exactness first).
        // The sweep is shared by the whole workgroup (a handful of angles, thousands of points): the angles publish themselves
        // through the 180 private dummy words of the real angles (never used otherwise), 45 at a time.
        const bool mine = tid < NUM_ANGLE && !C.in_lds;
        int n_out;
        const int rank = h3_scan(mine ? 1 : 0, n_out, fix + 32, lane, wave);
        unsigned* pub = cells + cell_dw;  // [45][4]: angle, w_lo, w_size, max
        int* const acc_base = P.accum + (size_t)slot * NUM_ANGLE * P.numrho + P.half;
        for (int r0 = 0; r0 < n_out; r0 += 45) {
            const int cnt = min(45, n_out - r0);
            if (mine && rank >= r0 && rank < r0 + cnt) {
                unsigned* e = pub + 4 * (rank - r0);
                e[0] = (unsigned)tid; e[1] = (unsigned)w_lo; e[2] = (unsigned)w_size; e[3] = 0;
            }
            __syncthreads();
            // Round 3: the bound of an angle is a HISTOGRAM of its window, and the LDS behind the placed windows has room for one
            // (16-bit counters, two per dword: what did not fit the windows of ALL angles fits one at a time): LDS atomics
            // instead of a sweep of global atomics, a wait for them, a read-back through the L2 and a sweep of stores to leave
            // the pooled accumulator all-zero (17-20 of a 4K streak frame's 44 us of set-up).  An angle whose window does not
            // fit even alone keeps the global sweep.
            unsigned* const hist = pub + HOUGH3_THREADS;
            const int hist_dw = (int)(P.lds_bytes / 4) - (int)(hist - lds);
            bool any_global = false;
            for (int k = 0; k < cnt; ++k) {
                const int a = (int)pub[4 * k], lo = (int)pub[4 * k + 1], size = (int)pub[4 * k + 2];
                const int need = (size + 1) >> 1;
                if (need > hist_dw) { any_global = true; continue; }  // (the same decision in every thread)
                const float ca = P.trig[2 * a], sa = P.trig[2 * a + 1];
                for (int d = tid; d < need; d += HOUGH3_THREADS) hist[d] = 0;
                __syncthreads();
                for (int p = tid; p < N; p += HOUGH3_THREADS) {
                    const unsigned pt = list[p];
                    const int r = __float2int_rn(__fadd_rn(__fmul_rn((float)(pt & 0xFFFF), ca), __fmul_rn((float)(pt >> 16), sa))) - lo;
                    atomicAdd(&hist[r >> 1], 1u << (16 * (r & 1)));
                }
                __syncthreads();
                unsigned m = 0;
                for (int d = tid; d < need; d += HOUGH3_THREADS) {
                    const unsigned v = hist[d];
                    m = max(m, max(v & 0xFFFFu, v >> 16));
                }
                m = wave_max_u32(m);
                if (lane == 0 && m) atomicMax(&pub[4 * k + 3], m);
                __syncthreads();
            }
            if (any_global) {
            for (int k = 0; k < cnt; ++k) {
                const int a = (int)pub[4 * k];
                if ((((int)pub[4 * k + 2] + 1) >> 1) <= hist_dw) continue;
                const float ca = P.trig[2 * a], sa = P.trig[2 * a + 1];
                int* row = acc_base + (size_t)a * P.numrho;
                for (int p = tid; p < N; p += HOUGH3_THREADS) {
                    const unsigned pt = list[p];
                    atomicAdd(&row[__float2int_rn(__fadd_rn(__fmul_rn((float)(pt & 0xFFFF), ca), __fmul_rn((float)(pt >> 16), sa)))], 1);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // every thread's atomics have been performed at the L2
            for (int k = 0; k < cnt; ++k) {
                const int a = (int)pub[4 * k], lo = (int)pub[4 * k + 1], size = (int)pub[4 * k + 2];
                if (((size + 1) >> 1) <= hist_dw) continue;
                const int* row = acc_base + (size_t)a * P.numrho;
                int m = 0;
                for (int r = tid; r < size; r += HOUGH3_THREADS) m = max(m, (int)ld_l2((const unsigned*)&row[lo + r]));
                m = (int)wave_max_u32((unsigned)m);
                if (lane == 0 && m) atomicMax(&pub[4 * k + 3], (unsigned)m);
            }
            }
            __syncthreads();
            if (mine && rank >= r0 && rank < r0 + cnt) C.dead = (int)pub[4 * (rank - r0) + 3] < P.threshold;
            if (any_global) {
            for (int k = 0; k < cnt; ++k) {  // back to all-zero: the replay (or the next frame in this slot) starts from there
                const int a = (int)pub[4 * k];
                if ((((int)pub[4 * k + 2] + 1) >> 1) <= hist_dw) continue;
                const float ca = P.trig[2 * a], sa = P.trig[2 * a + 1];
                int* row = acc_base + (size_t)a * P.numrho;
                for (int p = tid; p < N; p += HOUGH3_THREADS) {
                    const unsigned pt = list[p];
                    row[__float2int_rn(__fadd_rn(__fmul_rn((float)(pt & 0xFFFF), ca), __fmul_rn((float)(pt >> 16), sa)))] = 0;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
        }
        if (n_out > 0) {
            for (int k = tid; k < 180; k += HOUGH3_THREADS) pub[k] = 0;  // (the dummy words of the spare lanes, 180..191, were never touched)
            __threadfence();
            __syncthreads();
        }
    }
    long long dbg_alive = 0, dbg_global = 0;  // (MDH_HOUGH_DEBUG) angles on the pooled global accumulator: still in the replay / all
    if (TM && HYBRID) {  // (fix[26], fix[27]: spare words of the fixed block, zeroed at the frame's start)
        if (tid < NUM_ANGLE && !C.in_lds) atomicAdd(&fix[26], 1u);
        if (tid < NUM_ANGLE && !C.in_lds && !C.dead) atomicAdd(&fix[27], 1u);
        __syncthreads();
        dbg_global = fix[26];
        dbg_alive = fix[27];
    }
    const long long w1 = TM && P.dbg ? wall_clock64() : 0;
    if (P.acc_mode) h3_accumulate<HYBRID, TM>(C, N, list, w_lo, w_size);
    else h3_replay<HYBRID, TM>(C, N, list, perm);
    if (tid == 0) P.out_nraw[i] = C.found;
    if (TM && P.dbg && tid == 0) {  // MDH_HOUGH_DEBUG=1: points, tier, 100 MHz ticks (set-up, PPHT), segments, LDS cells
        long long* d = P.dbg + (size_t)i * 8;
        d[0] = N; d[1] = (HYBRID ? 2 : (P.lds_bytes == P.lds_small ? 0 : 1)) | (dbg_alive << 8) | (dbg_global << 16); d[2] = w1 - w0; d[3] = wall_clock64() - w1; d[4] = C.found;
        d[5] = total; d[6] = wall_clock64(); d[7] = w0;
        if (P.dbg_phase)
            for (int k = 0; k < 8; ++k) P.dbg_phase[(size_t)i * 8 + k] = C.ph[k];
    }
    if (HYBRID) {
        // leave the pooled buffers all-zero: every accumulator cell ever touched belongs to some point of the list
        __syncthreads();
        if (tid < NUM_ANGLE && !C.in_lds && !C.dead) {
            for (int p = 0; p < N; ++p) {
                const unsigned pt = list[p];
                C.acc_row[__float2int_rn(__fadd_rn(__fmul_rn((float)(pt & 0xFFFF), tc), __fmul_rn((float)(pt >> 16), ts)))] = 0;
            }
        }
        if (!mask_lds) {
            for (int e = tid; e < TW; e += HOUGH3_THREADS) {
                const int ry = e / wpr;
                const size_t g = (size_t)(by0 + ry) * P.WW + wlo + (e - ry * wpr);
                if (dst[g]) gmask[g] = 0;
            }
        }
        __threadfence();
    }
}

// tiers 0 / 1: a machine-sized grid walks the tier's frame list (a pass has thousands of frames and most of them carry no
// streak: one workgroup per frame would be thousands of workgroups that leave at once, each waiting for its LDS first)
template <bool TM>
__global__ void __launch_bounds__(HOUGH3_THREADS) k_hough3(HoughParams P) {
    __builtin_amdgcn_s_setprio(HOUGH_WAVE_PRIO);
    extern __shared__ __attribute__((aligned(16))) unsigned lds_dyn[];
    const int count = *P.fcount;
    if (P.fqueue == nullptr) {
        for (int j = blockIdx.x; j < count; j += gridDim.x) {
            hough3_frame<false, TM>(P, P.flist[j], 0, lds_dyn);
            __syncthreads();
        }
        return;
    }
    // Round 6: a work queue, as in k_hough3_big.  A 4000-frame pass at n = 25 has ~2800 frames with points for the 512 workgroups that
    // are resident at a time, 50-300 us each: dealt out by workgroup index the launch lasts as long as the unluckiest workgroup's
    // five or six frames; from a queue, as long as the average plus one frame.
    for (;;) {
        if (threadIdx.x == 0) lds_dyn[0] = (unsigned)atomicAdd(P.fqueue, 1);
        __syncthreads();
        const int q = (int)lds_dyn[0];
        __syncthreads();
        if (q >= count) break;
        hough3_frame<false, TM>(P, P.flist[q], 0, lds_dyn);
        __syncthreads();
    }
}
// tier 2: gridDim.x workgroups walk the list, each with its own slot of the pooled accumulator / mask
template <bool TM>
__global__ void __launch_bounds__(HOUGH3_THREADS) k_hough3_hybrid(HoughParams P) {
    __builtin_amdgcn_s_setprio(HOUGH_WAVE_PRIO);
    extern __shared__ __attribute__((aligned(16))) unsigned lds_dyn[];
    const int count = *P.fcount;
    for (int j = blockIdx.x; j < count; j += gridDim.x) {
        hough3_frame<true, TM>(P, P.flist[j], blockIdx.x, lds_dyn);
        __syncthreads();
    }
}

// Tiers 2 and 1 in ONE launch with a work queue (round 3).  Both want a whole CU's LDS, and a 4K pass has more such frames than
// the machine has CUs (164 hybrid + 188 large-LDS frames in a 1000-frame pass of the bench stream).  As two launches side by side
// they shared the CUs as the dispatcher saw fit: the large-LDS launch usually got there first, most hybrid frames -- the longest,
// 330 us against 150 -- started a round late, and the stage lasted 0.63 ms for 0.34 ms' worth of CU time.  Here every workgroup
// takes the next frame from one queue, hybrid frames first (longest first), so the short frames fill in behind the long ones.
template <bool TM>
__global__ void __launch_bounds__(HOUGH3_THREADS) k_hough3_big(HoughParams P) {
    __builtin_amdgcn_s_setprio(HOUGH_WAVE_PRIO);
    extern __shared__ __attribute__((aligned(16))) unsigned lds_dyn[];
    const int c2 = P.tier_count[2], c1 = P.tier_count[1];
    for (;;) {  // (the queue index travels through the first dynamic LDS word: the launch takes the whole 160 KB, a static word more would not fit)
        if (threadIdx.x == 0) lds_dyn[0] = (unsigned)atomicAdd(&P.tier_count[HOUGH_TIERS], 1);
        __syncthreads();
        const int q = (int)lds_dyn[0];
        __syncthreads();
        if (q < c2) hough3_frame<true, TM>(P, P.tier_list[2 * P.K + q], blockIdx.x, lds_dyn);
        else if (q - c2 < c1) hough3_frame<false, TM>(P, P.tier_list[1 * P.K + (q - c2)], 0, lds_dyn);
        else break;
        __syncthreads();
    }
}

// The per-frame protocol (one frame, results wanted at once): routing, replay (whatever the tier) and result packing of
// frame 0 in ONE launch -- a pass of one frame is a chain of dependent commands, and every command saved is ~5 us of latency.
struct SingleResult {  // the pass's result block in (pinned, device-visible) HOST memory: no copy command behind the kernel
    int *off, *nraw, *lit;
    double* gap;
    FrameScalars* fscal;
    int4* packed;
    const FrameScalars* d_fscal;  // the frame's scalars as the noise chain left them ...
    const ScalarState* d_state;   // ... or (no sigma was due for the frame: no noise chain ran) the detector's running scalars
    unsigned* seq_flag;           // host word the caller polls: set to `seq` once every result above is visible to the host
    unsigned seq;
    // clean-after-use: what k_pass_prepare / k_noise_finalize would zero in front of the slot's NEXT pass is zeroed behind this
    // one, after the results are out, so that a frame's chain starts with the window kernel
    int clean;                    // (row / frame lit counts and the bounding box are reached through HoughParams)
    unsigned* blockcnt;
    unsigned n_blockcnt;
    long long* sums;              // the slot's noise sums [0..1]: left at zero (a due frame on the direct path accumulates into them)
};
template <bool TM>
__global__ void __launch_bounds__(HOUGH3_THREADS) k_hough_single(HoughParams P, SingleResult R) {
    __builtin_amdgcn_s_setprio(HOUGH_WAVE_PRIO);
    extern __shared__ __attribute__((aligned(16))) unsigned lds_dyn[];
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int tier = hough_classify_frame(P, 0, tid, lds_dyn + 64);  // (band scratch: dynamic LDS nothing uses yet)
        if (tid == 0) lds_dyn[0] = (unsigned)(tier + 1);
    }
    __syncthreads();
    const int tier = (int)lds_dyn[0] - 1;
    __syncthreads();
    if (tier == 0 || tier == 1) hough3_frame<false, TM>(P, 0, 0, lds_dyn);
    else if (tier == 2) hough3_frame<true, TM>(P, 0, 0, lds_dyn);
    else if (tier == 3 && tid < 64) ppht_one_wave_frame(P, 0, 0, lds_dyn);
    __threadfence();
    __syncthreads();
    const int n_raw = (int)ld_l2((const unsigned*)&P.out_nraw[0]);
    const int cnt = min(n_raw, MAX_LINES);
    for (int q = tid; q < cnt; q += HOUGH3_THREADS) R.packed[q] = ((const int4*)P.out_lines)[q];
    if (tid == 0) {
        R.off[0] = 0;
        R.off[1] = cnt;
        R.nraw[0] = n_raw;
        R.lit[0] = P.litcount[0];
        R.gap[0] = P.out_gap[0];
        if (R.d_state) {
            const ScalarState st = *R.d_state;
            FrameScalars f;
            f.thr_f = st.thr_f; f.snr = st.ema_v; f.sigma = st.sigma; f.thr = st.thr; f.pad = 0;
            R.fscal[0] = f;
        } else {
            R.fscal[0] = R.d_fscal[0];
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0 && R.seq_flag) __hip_atomic_store(R.seq_flag, R.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (R.clean) {  // the host already has its results; the slot's counters are left as those of a fresh one
        // (dst and its word list stay: get_binary / visu may still ask for the frame; the slot's next pass wipes the listed words)
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (int e = tid; e < P.H; e += HOUGH3_THREADS) ((int*)P.rowcount)[e] = 0;
        if (tid == 0) ((int*)P.litcount)[0] = 0;
        if (tid < 4) ((int*)P.bbox)[tid] = 0;
        if (tid < 2 && R.sums) R.sums[tid] = 0;
        for (unsigned e = tid; e < R.n_blockcnt / 4; e += HOUGH3_THREADS) ((uint4*)R.blockcnt)[e] = z;
        for (unsigned e = (R.n_blockcnt & ~3u) + tid; e < R.n_blockcnt; e += HOUGH3_THREADS) R.blockcnt[e] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// The visiting order of cv::HoughLinesP for an N-point list (hough.cpp: idx = rng.uniform(0, count); swap-remove), as a
// table: perm_N[k] = row-major index of the k-th visited point.  rnd[] is the output sequence of cv::RNG(0xFFFFFFFFFFFFFFFF)
// (host).  One workgroup per N; the shuffle itself is a serial chain run by one lane on an LDS copy.
__global__ void __launch_bounds__(64) k_perm_build(const unsigned* __restrict__ rnd, unsigned short* __restrict__ perm, int n_max) {
    __shared__ unsigned short a[HOUGH_MAX_LIST], ix[HOUGH_MAX_LIST];
    const int N = blockIdx.x + 1;
    if (N > n_max) return;
    for (int k = threadIdx.x; k < N; k += 64) {
        a[k] = (unsigned short)k;
        ix[k] = (unsigned short)(rnd[k] % (unsigned)(N - k));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned short* out = perm + (size_t)N * (N - 1) / 2;
        for (int k = 0; k < N; ++k) {
            const int idx = ix[k];
            const unsigned short v = a[idx];
            a[idx] = a[N - 1 - k];
            out[k] = v;
        }
    }
}

#endif  // MDH_WINDOW_ONLY

// metdet_hip.hip -- host side of libmetdet_hip.so: the C-ABI of include/metdet_hip.h over the gfx950
// kernels in metdet_kernels.hpp.  No torch, no Triton, no CUDA-compat layer.
//
// Host logic mirrored from the reference (all data-independent, so it stays on the host):
//   * when the noise sigma is due          SNR_SW.update            Detector.py:81-91
//   * the EMA momentum warm-up schedule    EMA.adjust_weight        utils.py:360-368
//   * lineset_nms                          utils.py:780-839
// Everything that touches pixels runs on the device; the running scalars (snr, threshold) live on the device
// too, so a pass of K frames is queued without any host round trip.
//
// Streams.  A pass = pixel stages (noise, window/threshold, morphology, dynamic mask) on the MAIN stream, then the
// Hough stage on two HOUGH streams (small-LDS and big-LDS launches side by side).  Up to two passes are in flight:
// the pixel stages of pass p+1 (HBM-bound) overlap the Hough stage of pass p (LDS / latency-bound).  Buffers that
// cross the pixel -> Hough boundary and the result mirrors exist twice ("slots"); state that is carried from frame
// to frame (frame ring, ring of binaries, running scalars) is only touched on the main stream, in stream order.
#include "metdet_kernels.hpp"
#include "metdet_window.hpp"
#include "metdet_pre.hpp"
#include "metdet_host.hpp"  // error text, lineset_nms, the native collector: host-only code, also built alone under ASan / UBSan

#include "../../include/metdet_hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>


#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            set_err(std::string(#call) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" +            \
                    std::to_string(__LINE__) + ")");                                                   \
            return MDH_E_HIP;                                                                          \
        }                                                                                              \
    } while (0)

// Cross-stream ordering of a pass.  In "lean" mode (the per-frame protocol: one frame, results wanted at once) the whole pass is
// queued on ONE stream, so the waits and the events they wait for are dropped; only the completion event remains.
#define EV_WAIT(c, st, ev) do { if (!(c)->lean) HIPCHK(hipStreamWaitEvent(st, ev, 0)); } while (0)
#define EV_REC(c, ev, st) do { if (!(c)->lean) HIPCHK(hipEventRecord(ev, st)); } while (0)

enum KernelKind { KK_NOISE_SUMS = 0, KK_NOISE_FINAL, KK_STACK_THR, KK_MORPH, KK_DY_ALIVE, KK_DY_APPLY, KK_HOUGH, KK_CLASSIC };
static const char* kKernelNames[MDH_NUM_KERNELS] = {"noise_sums", "noise_finalize", "stack_threshold", "morph_close",
                                                    "dy_alive",   "dy_sparse",      "hough_ppht",      "classic"};

struct ProfSpan {
    int kind;
    hipEvent_t a, b;
};

#define NSLOT_MAX 8  // slots allocated at most; ctx->nslot (default 6, MDH_SLOTS) passes may be in flight
#define NHSET 4  // Hough buffer / stream sets available; MDH_HOUGH_SETS (default 2) of them are used
#define PACK_HEAD 8  // packed segments per frame (pass average) copied to the host without waiting for the counts

// Everything that crosses the pixel-stage -> Hough boundary or is mirrored to the host, per in-flight pass.
struct Slot {
    // pass header [pointer table (n-1+K) | noise jobs (K)]: one pinned block, one device block, ONE host-to-device copy
    uint8_t *h_hdr = nullptr, *d_hdr = nullptr;
    size_t hdr_jobs_off = 0;
    int nj = 0;                          // due frames of the pass
    const uint8_t** h_ptrtab = nullptr;  // view into h_hdr
    NoiseJob* h_jobs = nullptr;          // view into h_hdr
    // written on the staging / noise stream, read by the pixel stages of the same pass
    const uint8_t** d_ptrtab = nullptr;
    NoiseJob* d_jobs = nullptr;
    long long* d_sums = nullptr;
    ScalarState* d_jobres = nullptr;
    int* d_thr = nullptr;
    WinRec* d_wrec = nullptr;  // [K + WIN_PAD] the window kernel's per-frame records (k_noise_finalize)
    hipEvent_t ev_noise = nullptr;
    // dst is only written where it is non-zero; d_nzlist names the candidate words and k_sparse_clear zeroes them again
    // on the staging stream before the slot's next pass (k_pass_prepare, ahead of ev_noise).
    unsigned* d_bandflag = nullptr;  // [K][bands][WW]: set bits of b per (8-row band, word column) block
    unsigned *d_nzlist = nullptr, *d_nzcount = nullptr;
    ActWord* d_bbits = nullptr;      // [K][H][WW] b = (diff > T), sparse + tagged: written by the window kernel, read by the morphology
    u64* d_bplain = nullptr;         // [K][H][WW] the same image as plain words (dense passes; allocated when the first one is queued)
    bool dense = false;              // the pass in the slot runs in the dense format
    u64* d_planes = nullptr;         // Classic: [K][2][H][WW]
    ActWord* d_cplanes = nullptr;    // Classic, sparse format (round 6): [K][2][H][WW] tagged words, only those with set bits ever written
    bool classic_dense_last = false; // the slot's last Classic pass ran the dense kernels (they write every word of dst: a sparse pass clears it first)
    bool classic_sparse = false;     // format of the pass being queued
    unsigned* d_worklist = nullptr;  // [K][bands][WW] morphology blocks that can produce output
    hipEvent_t ev_k1 = nullptr;      // window kernel done (main stream) -> the post stream may start
    hipEvent_t ev_close = nullptr;   // binaries of this pass are in the ring (the next pass's dynamic mask reads them)
    u64* d_dst = nullptr;
    int *d_rowcount = nullptr, *d_litcount = nullptr, *d_bbox = nullptr;
    unsigned* d_workcnt = nullptr;   // [K] listed morphology blocks per frame (tail of the d_rowcount block)
    unsigned* d_dense = nullptr;     // [1 + K] count + list of the pass's dense frames (k_morph_close_dense's work), same block
    int* d_lines = nullptr;
    // result block [off (K+1) | nraw (K) | lit counts (K) | gap (K) | frame scalars (K) | packed segments]: same layout on
    // the device and in pinned host memory, so the scalars and the head of the packed segments travel in ONE copy
    uint8_t *d_res = nullptr, *h_res = nullptr;
    size_t res_lines_off = 0;
    int *d_off = nullptr, *h_off = nullptr, *d_nraw = nullptr, *h_nraw = nullptr, *d_litcopy = nullptr, *h_litcount = nullptr;
    double *d_gap = nullptr, *h_gap = nullptr;
    FrameScalars *d_fscal = nullptr, *h_fscal = nullptr;
    int *d_packed = nullptr, *h_lines = nullptr;  // segments packed back to back
    hipEvent_t ev_pix = nullptr, ev_done = nullptr;
    std::vector<ProfSpan> spans;
    // the pass occupying the slot
    bool busy = false, used = false;
    bool host_complete = false;  // the pass wrote its whole result block into host memory itself (k_hough_single)
    // per-frame protocol: k_hough_single zeroes behind itself what k_pass_prepare would zero in front of the slot's next pass
    // (lean_clean: the slot is in that state now; was_clean: it was when the pass being queued took it), and announces its
    // results through a word in the pinned result block that the host polls
    bool lean_clean = true, was_clean = false;
    bool was_lean = false;  // the slot's previous pass was a per-frame one (everything on the staging stream)
    unsigned* h_seq = nullptr;
    unsigned seq_no = 0;
    bool seq_armed = false;
    const uint8_t** h_hdr_dev = nullptr;  // device-visible address of the pinned header (the kernels read the table from host memory)
    int K = 0;
    int64_t t_base = 0;
};

// Hough-private buffers and streams.  Consecutive passes alternate between the sets, so the (latency-bound) Hough stage of
// one pass overlaps the next pass's instead of queueing behind it.
struct HoughSet {
    // [0] routing + small-LDS launch + result copies, [1] hybrid + one-wave launches (the pooled buffers), [2] large-LDS launch
    hipStream_t stream[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_big = nullptr, ev_big2 = nullptr, ev_cls = nullptr;
    // Pooled buffers of the frames that cannot live in LDS alone (hybrid accumulator / mask: `pool` slots; global point lists
    // of the one-wave kernel: `pool_list` slots).  A workgroup keeps one slot and walks through its share of the frames.
    int pool = 0, pool_list = 0;
    u64* d_work = nullptr;
    int* d_accum = nullptr;
    unsigned *d_nz_orig = nullptr, *d_nz_work = nullptr;
    unsigned short* d_perm_big = nullptr;  // [pool][HOUGH_BIG_LIST]: visiting-order table of the frame a hybrid workgroup is on, when it has more than HOUGH_MAX_LIST points
    int *d_tier_list = nullptr, *d_tier_count = nullptr;  // [HOUGH_TIERS][Kmax], [HOUGH_TIERS]: frame routing of the pass in the set
    int last_slot = -1;  // slot of the pass that used the set last
};

// Host memory a kernel can read (the per-frame protocol's direct path): pools handed out by mdh_host_alloc, buffers registered with
// mdh_host_register, and buffers the library registered itself because a caller kept handing over the same addresses (a loader
// that recycles a fixed ring of frame buffers: ProcessVideoLoader's shared RawArray, videoloader.py:696-701, 746-772).  One list
// per PROCESS under a mutex: a pool freed through any context disappears for all of them.
struct mdh_ctx;
struct HostPool {
    uintptr_t base, bytes, dev;
    int device;
    int kind;        // 0: mdh_host_alloc, 1: mdh_host_register, 2: registered by the library (auto), 3: mdh_host_register of memory the caller had pinned itself
    mdh_ctx* owner;  // the context that unregisters an auto entry when it is destroyed
};
static std::mutex g_pool_mutex;
static std::vector<HostPool> g_pools;
#define AUTO_REG_SEEN 32           // addresses remembered while deciding
#define AUTO_REG_HITS 3            // the third hand-over of an address registers it
#define AUTO_REG_MAX_BUFFERS 64    // cap per context
struct mdh_ctx {
    int device = 0;
    int role = MDH_ROLE_FULL;
    int H = 0, W = 0, WW = 0, n = 0, R = 0, Kmax = 0;
    size_t HW = 0, words = 0;
    mdh_cfg cfg{};
    hipStream_t mstream[2] = {nullptr, nullptr};    // main streams: window kernel of even / odd passes
    hipStream_t stream = nullptr;                   // the main stream of the pass being queued
    hipStream_t nstream = nullptr;                  // staging + noise / threshold chain (runs one pass ahead of the main stream)
    hipStream_t pstream[2] = {nullptr, nullptr};    // post streams: morphology + dynamic mask of even / odd passes
    HoughSet hset[NHSET];
    hipEvent_t ev_base = nullptr;  // MDH_HOST_LOG: origin of the GPU-side span times
    unsigned tag_lds = 0;  // MDH_TAG_PASS=1: dynamic LDS bytes that tag a pass's kernels in a rocprofv3 trace (1024 * (1 + pass % 3))
    int nhset = 1;  // (one set since the three-wave PPHT: the stage is short, and every extra stream costs the command processor)
    // ONE STREAM PER PASS (round 6; contexts made for short passes: max_batch x frame <= MDH_ONE_STREAM_MAX_MB, default 2500 MB = 1200 frames of
    // 1080p): a pass's window kernel, morphology, dynamic mask and Hough stage all go on ONE stream -- one of NCHAIN, pass after pass --
    // with a Hough buffer set per chain stream.  The by-stage streams above hand a pass from stream to stream four times, and every
    // hand-over (event record -> the other queue's barrier packet) costs 40-50 us of nothing: invisible behind a 1.5 ms window
    // kernel, but a 250-frame pass (window kernel 0.1 ms) lasted 0.6 ms from its first launch to its results, one such pass
    // completing every 0.23 ms (rocprofv3 kernel trace, gpurun_out/r06_12).  On one stream the stages follow each other at kernel-boundary
    // cost, and NCHAIN passes run side by side.
    bool win_missing = false;  // launch_wide asked the window translation units for an instantiation none of them holds (a build error: loud)
    bool one_stream = false;
    hipStream_t pass_stream[NHSET] = {nullptr, nullptr, nullptr, nullptr};
    int nmain = 2;  // main streams in use (MDH_MAIN_STREAMS): with 2, the window kernel of pass p+1 ramps up while that of pass p drains
    int64_t pass_seq = 0;
    hipStream_t cstream = nullptr;                  // strided copy of the segments at collect time
    int64_t t = 0;  // frames submitted (SlidingWindow.timer)
    // EMA schedule (host mirror of utils.py:334-368)
    double ema_m0 = 0, ema_warm = 0;
    int64_t ema_t = 0;
    double coef_a = 0, coef_b = 0;
    int numrho = 0;
    // device buffers used on the main stream only
    uint8_t* d_ring = nullptr;   // frames that arrive from the HOST: R slots by frame number, allocated at the first such pass
    uint8_t* d_hist = nullptr;   // frames submitted in device memory: only the window history (the last n-1 of a pass) is kept,
    int R_hist = 0;              //   (nslot+1)(n-1) slots handed out by a running counter (see retire_frames)
    int64_t hist_pos = 0;
    uint8_t* d_zero = nullptr;
    // aliases of the slot being queued (set by slot_begin)
    const uint8_t** d_ptrtab = nullptr;
    const uint8_t** h_ptrtab = nullptr;
    NoiseJob* d_jobs = nullptr;
    long long* d_sums = nullptr;
    ScalarState* d_jobres = nullptr;
    int* d_thr = nullptr;
    WinRec* d_wrec = nullptr;
    unsigned* d_bandflag = nullptr;
    ScalarState* d_state = nullptr;  // running scalars, only touched on the noise stream
    ActWord* d_bbits = nullptr;  // alias of the current slot's buffer (set by slot_begin)
    ActWord* d_act = nullptr;  // [R][H][WW] tagged sparse ring of closed binaries
    int prev_slot = -1;        // slot of the pass submitted before the one being queued
    float* d_trig = nullptr;
    const unsigned short* d_perm = nullptr;  // PPHT visiting-order tables (shared by every context on the device)
    const unsigned* d_rnd = nullptr;         // the first HOUGH_BIG_LIST outputs of cv::RNG(-1) (likewise)
    int lds_small = 80 * 1024, lds_large = 160 * 1024;  // dynamic LDS of the two three-wave PPHT launches (MDH_HOUGH_LDS_SMALL / _LARGE)
    bool hough_bands = true;                            // frames routed by their band windows where the corner windows miss the small launch (MDH_HOUGH_BANDS)
    int lds_tiny = 40 * 1024;                           // ... and of the launch for short streaks (MDH_HOUGH_LDS_TINY; 0: no such launch)
    long long* d_hdbg = nullptr;  // MDH_HOUGH_DEBUG=1: per-frame cycle stamps of the last pass
    uint8_t* d_expand = nullptr;
    Slot slot[NSLOT_MAX];
    int nslot = 6;
    int next_slot = 0;       // slot the next submitted pass takes
    int oldest = 0;          // slot of the oldest pass not yet collected
    int in_flight = 0;
    std::vector<const uint8_t*> hist;  // device pointers of the last n-1 frames, oldest first
    // row-band tiling: pass staged by mdh_band_noise, waiting for the global sums (mdh_band_pixels)
    Slot* band_slot = nullptr;
    int band_nj = 0;
    u64* d_xchg = nullptr;     // exchange staging (band: owned rows going out; Hough: pieces coming in from the host)
    size_t xchg_words = 0;
    // update()/detect() protocol
    bool pending = false;
    bool lean = false;         // the pass being queued runs on the staging stream alone (mdh_update / mdh_detect)
    hipEvent_t ev_copy = nullptr;  // lean mode: the caller's frame has been read
    // lean mode, the caller's buffer is pinned host memory and no sigma is due: the window kernel reads the frame (and the
    // pointer table) from host memory itself and leaves the ring copy; the chain is window -> morphology -> mask -> Hough
    const uint8_t* lean_direct = nullptr;  // device-visible address of the caller's frame, or null
    bool lean_direct_on = true;            // MDH_LEAN_DIRECT=0 switches the path off (A/B)
    bool lean_due = false;                 // the direct frame being queued is one whose noise sigma is due (a short noise chain in front)
    bool lean_due_on = true;               // MDH_LEAN_DUE=0: due frames take the general path (DMA copy, prepare, noise chain) as in round 2
    bool lean_poll = true;                 // MDH_LEAN_POLL=0: wait on the event instead of polling the result block
    // Window kernel launched ahead of its frame (k_stack_threshold_mail): queued at the end of mdh_update behind the frame's
    // last launch, resident and waiting on the mailbox when the next frame arrives; the next mdh_update writes the arguments
    // and the sequence number instead of launching.  Anything else that is about to use the GPU cancels it first.
    MailBox* h_mail = nullptr;             // pinned
    unsigned mail_seq = 0;                 // sequence number of the kernel waiting on the mailbox (mail_armed)
    bool mail_armed = false;
    bool mail_on = true;                   // MDH_LEAN_MAIL=0: every frame's window kernel is launched by its own mdh_update (A/B)
    int mail_n = 0;                        // window length the waiting kernel was instantiated for
    unsigned mail_grid = 0;                // its grid (workgroups)
    bool mail_used = false;                // the frame being queued went through the mailbox
    bool mail_next = false;                // the next frame's window kernel will be queued behind the frame being queued
    bool lean_entry = false;               // slot_begin is being called by mdh_update
    bool lean_commit = false;              // slot_commit is being called by mdh_update
    // buffers in pageable memory seen lately (address, hand-overs): a repeated address is a recycled buffer and gets registered
    struct { uintptr_t addr; int hits; } reg_seen[AUTO_REG_SEEN] = {};
    int reg_seen_next = 0;
    std::atomic<int> auto_registered{0};   // buffers this context registered itself (unregistered by mdh_destroy; another context's
                                           //   mdh_host_unregister may count one down: atomic)
    bool auto_register_on = false;         // opt-in: MDH_AUTO_REGISTER=1 / mdh_set_auto_register(ctx, 1) (the caller then owes mdh_host_unregister before a free)
    int64_t direct_frames = 0, copied_frames = 0;  // mdh_update: frames read in place by the window kernel / copied first
    int64_t mail_hits = 0, mail_misses = 0;  // frames that found the waiting kernel alive / launches cancelled or expired
    double t_detect_ret = 0;               // host time mdh_detect last returned at; the gap to the next mdh_update decides whether the
    double mail_gap_us = 110;              //   caller is in a tight loop (MDH_LEAN_MAIL_GAP: a waiting kernel spins, so only then).  110 us
                                           //   (round 3: 40): a loader that refills a ring buffer between frames (a 2 MB copy: 60-100 us) still counts;
                                           //   the kernel's patience is 150 us (MAIL_EXPIRE_TICKS)
    bool tight_loop = false;
    // time-window sharding on a full context: noise sums of due frames supplied by the caller (the frames of a window that
    // restarted empty cannot produce them: mdh_inject_sums), and frames whose Hough stage is skipped (mdh_mute_hough_until)
    std::vector<std::pair<int64_t, std::pair<int64_t, int64_t>>> injected;  // (frame number, (S1, S2)), any order
    long long* h_inject = nullptr;         // pinned staging of the injected sums of the pass being queued
    int64_t mute_until = 0;
    int64_t seek_at = -1;  // mdh_seek to a position > 0: frames up to seek_at + n - 1 have windows that reach in front of it (k_noise_finalize's mute_pix)
    unsigned* d_read_count = nullptr;      // waves of the direct window kernel that have finished (zero between launches)
    unsigned* h_read_flag = nullptr;       // pinned word: read_seq once the caller's frame has been read
    unsigned read_seq = 0;
    // A pass mutates host-side stream state (EMA schedule, history, ring positions) while it is being queued; if a HIP call
    // fails half-way the context no longer describes the stream, so it refuses further work instead of producing silently
    // wrong thresholds / windows.
    std::vector<hipEvent_t> chain_ev;  // mdh_chain: a ring of events
    size_t chain_next = 0;
    bool poisoned = false;
    double wait_timeout_ms = 0;  // mdh_set_wait_timeout_ms: how long a collect waits for its pass (0 = for ever)
    bool planes_begun = false;      // MDH_ROLE_HOUGH: the next slot's counters are zeroed and being filled by mdh_band_pixels_submit_to
    int planes_K = 0;               //   ... with this many frames per band (every band of the pass, and the submit, must agree)
    hipEvent_t ev_begin = nullptr;  //   ... that zeroing, for the bands' post streams to wait on
    bool band_thr_done = false;     // mdh_band_thresholds_submit ran for the pass staged by mdh_band_submit
    bool streams_borrowed = false;  // mdh_share_streams: the main / noise / post / copy streams belong to another context
    // Format of b = (diff > T) between the window kernel and the morphology: sparse + tagged (a quiet sky: nothing is written for
    // 99 % of the wave-frames) or dense + plain (binaries with noise in them: 8K at m3det_high).  Chosen per pass from the share
    // of words with set bits the last collected pass reported (both formats report it); identical results either way.
    int dense_cfg = -1;        // MDH_DENSE: 1 = always dense, 0 = never, unset = automatic
    bool dense_now = false;
    // ClassicDetector's output format (round 6): sparse (tagged difference planes, block lists) unless MDH_CLASSIC_SPARSE=0; a pass
    // that lists more than a quarter of its blocks (a lowered threshold lets the noise through) sends the next 32 passes to the dense kernels
    bool classic_sparse_on = true;
    bool classic_fast4 = true;  // MDH_CLASSIC_FAST4=0: the general time-blocked kernel with sparse stores (A/B and parity)
    int classic_dense_passes = 0;
    double dense_leave = 0.015;  // back to the sparse format below this share of words with set bits (MDH_DENSE_LEAVE)
    int64_t dense_passes = 0;  // passes queued in the dense format (mdh_dense_passes)
    // hand-over between run_pixels and the window kernel's launcher: dense output wanted (and where), and whether the kernel
    // that was launched honours it (only the wide-load variants do; the others fall back to the sparse format)
    bool win_dense = false, win_dense_used = false;
    u64* win_plain = nullptr;
    unsigned long long* win_nzwords = nullptr;
    double last_density = 0;
    int last_slot = -1;        // slot of the most recently collected pass (mdh_get_binary)
    // switches
    bool force_generic = false;     // MDH_FORCE_GENERIC=1: use the n-reads kernel (A/B and parity testing)
    bool force_bytes = false;       // MDH_FORCE_BYTES=1: never use the wide-load window kernel
    bool force_ppl8 = false;        // MDH_FORCE_PPL8=1: 8 pixels per lane even when 16 would do
    bool hough_serial = false;      // MDH_HOUGH_SERIAL=1: one-point-at-a-time PPHT replay
    bool hough_global_acc = false;  // MDH_HOUGH_GLOBAL_ACC=1: never use the LDS accumulator windows
    bool hough_one_wave = false;    // MDH_HOUGH_ONE_WAVE=1: one wave per frame instead of a 256-thread workgroup
    // profiling
    bool prof = false;         // timing events around the kernels of the pass being queued
    int prof_every = 0;        // 0: off; N: every N-th pass is timed (mdh_profile_enable)
    std::vector<ProfSpan>* cur_spans = nullptr;
    std::vector<hipEvent_t> ev_pool;  // recycled timing events
    double host_us[6] = {0};  // MDH_HOST_TRACE=1: submit, wait for ev_done, segment copy, nms + unpack, (spare)
    int64_t host_n = 0;
    double prof_ms[MDH_NUM_KERNELS] = {0};
    int64_t prof_n[MDH_NUM_KERNELS] = {0};
};

// GPU_MAX_HW_QUEUES as the process environment has it (0: unset, the runtime's default of 4 applies); the library wants >= 12
extern "C" int mdh_hw_queues_hint(void) {
    const char* v = getenv("GPU_MAX_HW_QUEUES");
    return v ? atoi(v) : 0;
}
// hipSetDevice for host code that has no HIP binding of its own (metdetpy_amd/rccl.py selects the device before ncclCommInitRank)
extern "C" int mdh_set_device(int device) {
    HIPCHK(hipSetDevice(device));
    return MDH_OK;
}
// Context-free device helpers for a host's own collectives (metdetpy_amd/dist.py: the final detection gather is an ncclAllGather on a
// stream of the host's, staged through a few bytes of device memory -- no mdh_ctx involved, no torch tensor either).
extern "C" int mdh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}
extern "C" void* mdh_stream_create(int device) {
    hipStream_t s = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
        set_err("mdh_stream_create: hipStreamCreateWithFlags failed");
        return nullptr;
    }
    return (void*)s;
}
extern "C" int mdh_stream_destroy(int device, void* stream) {
    if (!stream) return MDH_E_ARG;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamDestroy((hipStream_t)stream));
    return MDH_OK;
}
// waits for everything queued on `stream`; timeout_ms > 0: gives up with MDH_E_TIMEOUT (a collective whose peer never arrived)
extern "C" int mdh_stream_sync(int device, void* stream, double timeout_ms) {
    HIPCHK(hipSetDevice(device));
    if (timeout_ms <= 0) {
        HIPCHK(hipStreamSynchronize((hipStream_t)stream));
        return MDH_OK;
    }
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        hipError_t e = hipStreamQuery((hipStream_t)stream);
        if (e == hipSuccess) return MDH_OK;
        if (e != hipErrorNotReady) { set_err(std::string("mdh_stream_sync: ") + hipGetErrorString(e)); return MDH_E_HIP; }
        timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6 > timeout_ms) {
            set_err("mdh_stream_sync: the stream's work did not complete in time");
            return MDH_E_TIMEOUT;
        }
        timespec nap = {0, 50000};
        nanosleep(&nap, nullptr);
    }
}
extern "C" void* mdh_dev_malloc(int device, uint64_t bytes) {
    void* p = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) {
        set_err("mdh_dev_malloc: hipMalloc failed");
        return nullptr;
    }
    return p;
}
extern "C" int mdh_dev_free(int device, void* p) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipFree(p));
    return MDH_OK;
}
// to_device = 1: host -> device, 0: device -> host; queued on `stream` and waited for (small control buffers)
extern "C" int mdh_dev_copy(int device, void* dst, const void* src, uint64_t bytes, int to_device, void* stream) {
    if (!dst || !src) return MDH_E_ARG;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return MDH_OK;
}
extern "C" const char* mdh_kernel_name(int k) { return (k >= 0 && k < MDH_NUM_KERNELS) ? kKernelNames[k] : ""; }
extern "C" int64_t mdh_frames_seen(const mdh_ctx* c) { return c ? c->t : -1; }
extern "C" int mdh_in_flight(const mdh_ctx* c) { return c ? c->in_flight : -1; }
extern "C" int mdh_max_in_flight(const mdh_ctx* c) { return c ? c->nslot : -1; }
extern "C" int64_t mdh_dense_passes(const mdh_ctx* c) { return c ? c->dense_passes : -1; }
extern "C" double mdh_last_density(const mdh_ctx* c) { return c ? c->last_density : -1; }
extern "C" int mdh_dense_now(const mdh_ctx* c) {
    if (!c) return -1;
    if (c->cfg.detector != MDH_DET_M3) return (!c->classic_sparse_on || c->classic_dense_passes > 0) ? 1 : 0;  // ClassicDetector: its own two formats
    return c->dense_now ? 1 : 0;
}

// Dense passes: where do the morphology's candidate blocks come from (MDH_DENSE_CAND)?
//   1 (default)  the window kernel flags them itself, with the neighbours a pair at a block's edge can reach: the selection lists
//                exactly the flagged blocks, the plain words are read back only there;
//   2            a streaming sweep over the plain words flags them (k_dense_candidates), the selection dilates by one block (the
//                first form of round 5: the sweep reads all 2 GB back -- A/B and parity);
//   0            no candidates: every word of every frame through k_morph_close_plain (round 4 -- A/B and parity).
static int dense_cand_mode() {
    static const int mode = getenv("MDH_DENSE_CAND") ? atoi(getenv("MDH_DENSE_CAND")) : 1;
    return (mode == 1 || mode == 3) ? (WIDE_DENSE_CAND != 0 ? 1 : 2) : (mode == 2 ? 2 : 0);
}
static bool dense_candidates_inline() { return dense_cand_mode() == 1; }
static bool dense_candidates(const mdh_ctx*) { return dense_cand_mode() != 0; }

template <typename T>
static int dev_alloc(T*& p, size_t count, bool zero = true) {
    HIPCHK(hipMalloc((void**)&p, count * sizeof(T)));
    if (zero) HIPCHK(hipMemset(p, 0, count * sizeof(T)));
    return MDH_OK;
}
template <typename T>
static int host_alloc(T*& p, size_t count) {
    HIPCHK(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault));
    memset(p, 0, count * sizeof(T));
    return MDH_OK;
}

static inline double now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
static bool g_host_log = getenv("MDH_HOST_LOG") != nullptr;  // per-call host timestamps (three clocks, to line up with a trace)
static void host_log(const char* what, long long seq) {
    if (!g_host_log) return;
    timespec a, b, c;
    clock_gettime(CLOCK_MONOTONIC, &a);
    clock_gettime(CLOCK_MONOTONIC_RAW, &b);
    clock_gettime(CLOCK_BOOTTIME, &c);
    fprintf(stderr, "[mdh log] %-14s seq %lld  mono %lld  raw %lld  boot %lld\n", what, seq, a.tv_sec * 1000000000LL + a.tv_nsec,
            b.tv_sec * 1000000000LL + b.tv_nsec, c.tv_sec * 1000000000LL + c.tv_nsec);
}

static void free_all(mdh_ctx* c) {
    if (!c) return;
    if (getenv("MDH_HOST_TRACE") && c->host_n)
        fprintf(stderr, "[mdh host] passes %lld  per pass: submit %.1f us, wait %.1f us, segment copy %.1f us, nms+unpack %.1f us\n",
                (long long)c->host_n, c->host_us[0] / c->host_n, c->host_us[1] / c->host_n, c->host_us[2] / c->host_n,
                c->host_us[3] / c->host_n);
    hipSetDevice(c->device);
    if (c->mail_armed && c->h_mail) __atomic_store_n(&c->h_mail->line[0][15], ~c->mail_seq, __ATOMIC_RELEASE);  // (mail_cancel is declared further down)
    if (getenv("MDH_HOST_TRACE") && (c->mail_hits || c->mail_misses))
        fprintf(stderr, "[mdh host] window kernels launched ahead of their frame: %lld took their frame, %lld cancelled / expired\n",
                (long long)c->mail_hits, (long long)c->mail_misses);
    hipDeviceSynchronize();
    {   // buffers this context registered on its callers' behalf
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (size_t k = 0; k < g_pools.size();) {
            if (g_pools[k].owner == c && g_pools[k].kind == 2) {
                if (hipHostUnregister((void*)g_pools[k].base) != hipSuccess) (void)hipGetLastError();  // (the caller may have freed it already)
                g_pools.erase(g_pools.begin() + k);
            } else {
                if (g_pools[k].owner == c) g_pools[k].owner = nullptr;
                ++k;
            }
        }
    }
    void* dev[] = {c->d_ring, c->d_hist, c->d_zero, c->d_state, c->d_act, c->d_trig, c->d_hdbg, c->d_expand, c->d_xchg, c->d_read_count};
    for (void* p : dev)
        if (p) hipFree(p);
    if (c->h_read_flag) hipHostFree(c->h_read_flag);
    if (c->h_mail) hipHostFree(c->h_mail);
    if (c->h_inject) hipHostFree(c->h_inject);
    for (Slot& s : c->slot) {
        void* sdev[] = {s.d_bbits, s.d_bplain, s.d_planes, s.d_worklist, s.d_res, s.d_bandflag, s.d_nzlist, s.d_nzcount, s.d_dst, s.d_rowcount, s.d_lines, s.d_hdr, s.d_sums, s.d_jobres, s.d_thr, s.d_wrec, s.d_cplanes};
        for (void* p : sdev)
            if (p) hipFree(p);
        void* shost[] = {s.h_hdr, s.h_res};
        for (void* p : shost)
            if (p) hipHostFree(p);
        if (s.ev_noise) hipEventDestroy(s.ev_noise);
        if (s.ev_k1) hipEventDestroy(s.ev_k1);
        if (s.ev_close) hipEventDestroy(s.ev_close);
        if (s.ev_pix) hipEventDestroy(s.ev_pix);
        if (s.ev_done) hipEventDestroy(s.ev_done);
        for (auto& sp : s.spans) {
            hipEventDestroy(sp.a);
            hipEventDestroy(sp.b);
        }
    }
    for (hipEvent_t e : c->ev_pool) hipEventDestroy(e);
    if (c->ev_begin) hipEventDestroy(c->ev_begin);
    for (hipEvent_t e : c->chain_ev)
        if (e) hipEventDestroy(e);
    if (c->ev_copy) hipEventDestroy(c->ev_copy);
    for (HoughSet& h : c->hset) {
        void* hdev[] = {h.d_work, h.d_accum, h.d_nz_orig, h.d_nz_work, h.d_tier_list, h.d_tier_count, h.d_perm_big};
        for (void* p : hdev)
            if (p) hipFree(p);
        if (h.ev_big) hipEventDestroy(h.ev_big);
        if (h.ev_big2) hipEventDestroy(h.ev_big2);
        if (h.ev_cls) hipEventDestroy(h.ev_cls);
        for (hipStream_t q : h.stream)
            if (q) hipStreamDestroy(q);
    }
    hipStream_t st[] = {c->mstream[0], c->mstream[1], c->nstream, c->cstream, c->pstream[0], c->pstream[1]};
    for (hipStream_t q : st)
        if (q && !c->streams_borrowed) hipStreamDestroy(q);
    for (hipStream_t q : c->pass_stream)
        if (q) hipStreamDestroy(q);
    delete c;
}

// PPHT visiting-order tables (k_perm_build): a pure function of cv::RNG's fixed seed, built once per device and process
// (67 MB for lists of up to HOUGH_MAX_LIST points) and shared by every context.
static std::mutex g_perm_mutex;
static unsigned short* g_perm[64] = {nullptr};
static unsigned* g_rnd[64] = {nullptr};
static int perm_table(int device, const unsigned short** out, const unsigned** rnd_out) {
    std::lock_guard<std::mutex> lock(g_perm_mutex);
    if (device < 0 || device >= 64) { set_err("perm_table: device index out of range"); return MDH_E_ARG; }
    if (!g_perm[device]) {
        const int n_max = HOUGH_MAX_LIST;
        // (HOUGH_BIG_LIST entries of the generator's output stay on the device: a frame with more than HOUGH_MAX_LIST points builds
        // its own visiting-order table from them, hough3_frame)
        std::vector<unsigned> rnd(HOUGH_BIG_LIST);
        uint64_t st = 0xFFFFFFFFFFFFFFFFull;  // cv::RNG rng((uint64)-1), hough.cpp
        for (int k = 0; k < HOUGH_BIG_LIST; ++k) {
            st = (uint64_t)(unsigned)st * 4164903690ull + (st >> 32);  // RNG::next(): multiply-with-carry
            rnd[k] = (unsigned)st;
        }
        // built on a private stream and waited for with a stream synchronisation: other contexts' passes in flight on this
        // device are not stalled (a device-wide synchronisation would), and the temporaries are released on every path
        unsigned* d_rnd = nullptr;
        unsigned short* d_perm = nullptr;
        hipStream_t st_build = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&st_build, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void**)&d_rnd, sizeof(unsigned) * HOUGH_BIG_LIST);
        if (e == hipSuccess) e = hipMalloc((void**)&d_perm, sizeof(unsigned short) * ((size_t)n_max * (n_max + 1) / 2));
        if (e == hipSuccess) e = hipMemcpyAsync(d_rnd, rnd.data(), sizeof(unsigned) * HOUGH_BIG_LIST, hipMemcpyHostToDevice, st_build);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_perm_build, dim3(n_max), dim3(64), 0, st_build, d_rnd, d_perm, n_max);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st_build);
        if (st_build) hipStreamDestroy(st_build);
        if (e != hipSuccess) {
            if (d_perm) hipFree(d_perm);
            if (d_rnd) hipFree(d_rnd);
            set_err(std::string("perm_table: ") + hipGetErrorString(e));
            return MDH_E_HIP;
        }
        g_perm[device] = d_perm;
        g_rnd[device] = d_rnd;
    }
    *out = g_perm[device];
    *rnd_out = g_rnd[device];
    return MDH_OK;
}

// hipDeviceReset: everything this PROCESS holds on the device goes -- contexts (which must have been destroyed), the visiting-order
// tables, the runtime's hardware queues.  For a host that is done with the GPU but stays alive while other processes use it
// (bench.py before it runs its secondary workloads in subprocesses: an idle parent's queues still count when the command
// processor decides whether the device is oversubscribed and starts time-slicing).
extern "C" int mdh_device_reset(int device) {
    HIPCHK(hipSetDevice(device));
    {
        std::lock_guard<std::mutex> lock(g_perm_mutex);
        if (device >= 0 && device < 64) { g_perm[device] = nullptr; g_rnd[device] = nullptr; }  // (freed by the reset)
    }
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (size_t k = 0; k < g_pools.size();)
            if (g_pools[k].device == device) g_pools.erase(g_pools.begin() + k); else ++k;
    }
    HIPCHK(hipDeviceReset());
    return MDH_OK;
}

static int create_impl(mdh_ctx* c) {
    HIPCHK(hipSetDevice(c->device));
    // Stream priorities (experiment switches): MDH_PRIO_{MAIN,NOISE,HOUGH} = -1 low, 0 normal (default), 1 high.
    int prio_low = 0, prio_high = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    auto prio_of = [&](const char* name, int dflt) {
        const char* v = getenv(name);
        const int k = v ? atoi(v) : dflt;
        return k > 0 ? prio_high : (k < 0 ? prio_low : (prio_low + prio_high) / 2);
    };
    const bool use_prio = getenv("MDH_PRIO_MAIN") || getenv("MDH_PRIO_NOISE") || getenv("MDH_PRIO_HOUGH");
    // CU partition.  The window kernel's workgroups live for hundreds of frames and fill every CU, so the short,
    // latency-bound stages of the other passes in flight (morphology, dynamic mask, Hough: a serial chain per frame)
    // otherwise queue for a free wave slot instead of overlapping.  MDH_POST_CUS = m reserves m CUs per XCD for them
    // (experiment switch, default 0 = no partition; measured: no gain, the pipeline was
    // bound by stream dependencies, not by CU availability): the window streams get the complement.  Bit i of a HIP CU mask is CU i of the
    // device, and the runtimes differ in whether consecutive bits walk along one XCD or across the 8 XCDs, so the
    // reserved set {32x + x + 8j : x < 8, j < m} is chosen to put m CUs on every XCD under both numberings.
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, c->device));
    int post_cus = 0;
    if (const char* v = getenv("MDH_POST_CUS")) post_cus = atoi(v);
    const int ncu = prop.multiProcessorCount;
    if (ncu != 256 || post_cus < 0 || post_cus > 3 * 4 || c->role != MDH_ROLE_FULL) post_cus = 0;  // the set above assumes 8 XCDs x 32 CUs
    uint32_t mask_post[8] = {0}, mask_main[8];
    for (int x = 0; x < 8; ++x)
        for (int j = 0; j < post_cus; ++j) {
            const int bit = 32 * x + ((x + 8 * j) & 31);
            mask_post[bit >> 5] |= 1u << (bit & 31);
        }
    for (int k = 0; k < 8; ++k) mask_main[k] = ~mask_post[k];
    auto make_stream = [&](hipStream_t* st, const char* name) -> hipError_t {
        const bool is_main = strcmp(name, "MDH_PRIO_MAIN") == 0;
        if (post_cus > 0 && (is_main || getenv("MDH_POST_MASKED"))) return hipExtStreamCreateWithCUMask(st, 8, is_main ? mask_main : mask_post);
        return use_prio ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio_of(name, 0))
                        : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    };
    HIPCHK(make_stream(&c->mstream[0], "MDH_PRIO_MAIN"));
    HIPCHK(make_stream(&c->mstream[1], "MDH_PRIO_MAIN"));
    c->stream = c->mstream[0];
    HIPCHK(make_stream(&c->nstream, "MDH_PRIO_NOISE"));
    HIPCHK(make_stream(&c->pstream[0], "MDH_PRIO_HOUGH"));
    HIPCHK(make_stream(&c->pstream[1], "MDH_PRIO_HOUGH"));
    if (c->one_stream)
        for (hipStream_t& q : c->pass_stream) HIPCHK(make_stream(&q, "MDH_PRIO_MAIN"));
    HIPCHK(hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
    // MDH_HOUGH_CUS = m (experiment switch, default 0 = everywhere): the Hough streams alone are confined to m CUs of every XCD,
    // every other stream -- the window kernel's too -- keeps the whole machine.  (MDH_POST_CUS above takes the reserved CUs AWAY
    // from the window kernel; this one only keeps the Hough workgroups, which cost a CU a window workgroup while they are
    // resident, off most of the CUs.)
    int hough_cus = 0;
    if (const char* v = getenv("MDH_HOUGH_CUS")) hough_cus = atoi(v);
    if (ncu != 256 || hough_cus < 1 || hough_cus > 31) hough_cus = 0;
    uint32_t mask_hough[8] = {0};
    for (int x = 0; x < 8; ++x)
        for (int j = 0; j < hough_cus; ++j) {
            const int bit = 32 * x + ((x + 8 * j + 4 * (j / 4)) & 31);
            mask_hough[bit >> 5] |= 1u << (bit & 31);
        }
    auto make_hough_stream = [&](hipStream_t* st) -> hipError_t {
        if (hough_cus > 0) return hipExtStreamCreateWithCUMask(st, 8, mask_hough);
        return make_stream(st, "MDH_PRIO_HOUGH");
    };
    for (int hi = 0; hi < c->nhset; ++hi) {
        HoughSet& h = c->hset[hi];
        HIPCHK(make_hough_stream(&h.stream[0]));
        HIPCHK(make_hough_stream(&h.stream[1]));
        HIPCHK(make_hough_stream(&h.stream[2]));
        HIPCHK(hipEventCreateWithFlags(&h.ev_big, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h.ev_big2, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&h.ev_cls, hipEventDisableTiming));
    }
    const size_t K = c->Kmax, HW = c->HW, words = c->words;
    const size_t nb = (size_t)(c->H + MORPH_RB - 1) / MORPH_RB;
    int rc;
#define TRY(x) if ((rc = (x)) != MDH_OK) return rc
    // roles (row-band tiling): a BAND context runs no Hough stage, a HOUGH context no pixel stages
    const bool pix = c->role != MDH_ROLE_HOUGH, hough = c->role != MDH_ROLE_BAND;
    if (pix) {
        c->R_hist = (c->nslot + 1) * (c->n - 1);
        if (c->R_hist) TRY(dev_alloc(c->d_hist, (size_t)c->R_hist * HW, false));
        TRY(dev_alloc(c->d_zero, HW));
        if (c->cfg.detector == MDH_DET_M3) TRY(dev_alloc(c->d_act, (size_t)c->R * words));
    }
    TRY(dev_alloc(c->d_state, 1));
    TRY(dev_alloc(c->d_read_count, 64 + MAIL_PAYLOAD + 8));  // [0] finished waves of the direct window kernel, [32] the mailbox kernel's go word, [64..] its payload
    TRY(host_alloc(c->h_mail, 1));
    TRY(host_alloc(c->h_read_flag, 16));
    c->h_read_flag[0] = 0;
    if (pix) TRY(host_alloc(c->h_inject, (size_t)2 * 64 * NSLOT_MAX));  // staging of mdh_inject_sums entries (at most 64 pending)
    for (int hi = 0; hi < c->nhset && hough; ++hi) {
        HoughSet& h = c->hset[hi];
        h.pool = (int)std::min<size_t>(K, 256);  // one hybrid workgroup per CU (it takes the whole LDS)
        // global point lists of the one-wave kernel (frames with more than 8192 lit pixels: a cloud edge): two lists of H*W entries per
        // slot.  Round 4 kept 16 slots: a pass with 640 such frames replayed them 16 at a time.  Now as many as 1 GiB pays for, at
        // least 16 and at most one per pooled accumulator (1080p: 64 slots, 4K and 8K: 16) -- per CONTEXT: a process that runs eight
        // of them next to 13 GB of frame ring each must still fit (4 GiB here made `bench.py --mode timeshard --host-input` run out of
        // memory); and since round 5 most such frames take the hybrid three-wave replay anyway (hough_classify_frame).
        h.pool_list = (int)std::min<size_t>(std::min<size_t>(K, (size_t)h.pool), std::max<size_t>(16, std::min<size_t>(256, ((size_t)1 << 30) / (8 * HW))));
        TRY(dev_alloc(h.d_work, (size_t)h.pool * words));
        TRY(dev_alloc(h.d_accum, (size_t)h.pool * NUM_ANGLE * c->numrho));
        TRY(dev_alloc(h.d_nz_orig, (size_t)h.pool_list * HW, false));
        TRY(dev_alloc(h.d_nz_work, (size_t)h.pool_list * HW, false));
        TRY(dev_alloc(h.d_perm_big, (size_t)h.pool * HOUGH_BIG_LIST, false));
        TRY(dev_alloc(h.d_tier_list, (size_t)HOUGH_TIERS * K));
        TRY(dev_alloc(h.d_tier_count, HOUGH_TIERS + 4));  // (+ the work queue's head of k_hough3_big)
    }
    if (hough) TRY(perm_table(c->device, &c->d_perm, &c->d_rnd));
    TRY(dev_alloc(c->d_trig, 2 * NUM_ANGLE));
    TRY(dev_alloc(c->d_expand, HW));
    if (getenv("MDH_HOUGH_DEBUG")) TRY(dev_alloc(c->d_hdbg, K * 16));  // [K][8] stamps, then [K][8] phase cycle totals
    for (int si = 0; si < c->nslot; ++si) {
        Slot& s = c->slot[si];
        {
            // (+ WIN_PAD copies of the last entry: the batched window kernels read their table without clamping the index, see WinRec)
            const size_t ptr_bytes = (sizeof(uint8_t*) * (c->n - 1 + K + WIN_PAD) + 15) & ~(size_t)15;
            s.hdr_jobs_off = ptr_bytes;
            TRY(host_alloc(s.h_hdr, ptr_bytes + sizeof(NoiseJob) * K));
            TRY(dev_alloc(s.d_hdr, ptr_bytes + sizeof(NoiseJob) * K));
            s.h_ptrtab = (const uint8_t**)s.h_hdr;
            s.h_jobs = (NoiseJob*)(s.h_hdr + ptr_bytes);
            s.d_ptrtab = (const uint8_t**)s.d_hdr;
            s.d_jobs = (NoiseJob*)(s.d_hdr + ptr_bytes);
            void* dv = nullptr;
            if (hipHostGetDevicePointer(&dv, s.h_hdr, 0) == hipSuccess) s.h_hdr_dev = (const uint8_t**)dv;
            else (void)hipGetLastError();
        }
        TRY(dev_alloc(s.d_sums, 2 * K));
        TRY(dev_alloc(s.d_jobres, K + 1));
        TRY(dev_alloc(s.d_thr, K));
        TRY(dev_alloc(s.d_wrec, K + WIN_PAD));
        HIPCHK(hipEventCreateWithFlags(&s.ev_noise, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s.ev_k1, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s.ev_close, hipEventDisableTiming));
        if (pix) {
            if (c->cfg.detector == MDH_DET_M3) {
                TRY(dev_alloc(s.d_bbits, K * words));
                TRY(dev_alloc(s.d_worklist, K * nb * c->WW, false));
            } else {
                TRY(dev_alloc(s.d_planes, 2 * K * words));
                if (c->classic_sparse_on) {
                    TRY(dev_alloc(s.d_cplanes, 2 * K * words));
                    TRY(dev_alloc(s.d_worklist, K * nb * c->WW, false));
                }
            }
            // [K][nb * WW] block bit counts | [K][ceil(nb * WW / 32)] bitmap of the blocks with a non-zero count (note_block)
            TRY(dev_alloc(s.d_bandflag, K * nb * c->WW + K * ((nb * c->WW + 31) / 32)));
            TRY(dev_alloc(s.d_nzlist, K * words, false));
        }
        TRY(dev_alloc(s.d_nzcount, 4));  // [0] listed words, [1] listed morphology blocks, [2..3] u64: words of b with set bits (window kernel)
        TRY(dev_alloc(s.d_dst, K * words));
        // one block [rowcount K*H | litcount K | bbox 4K] so that a single memset clears all three per pass
        // (+ K: the per-frame counts of the morphology work list, d_workcnt, cleared with the rest)
        TRY(dev_alloc(s.d_rowcount, K * c->H + (6 + WORKCNT_STRIDE) * K + 1));
        s.d_litcount = s.d_rowcount + K * c->H;
        s.d_bbox = s.d_litcount + K;
        s.d_workcnt = (unsigned*)(s.d_bbox + 4 * K);
        s.d_dense = s.d_workcnt + (size_t)WORKCNT_STRIDE * K;
        TRY(dev_alloc(s.d_lines, K * (size_t)MAX_LINES * 4));
        {
            auto a16 = [](size_t b) { return (b + 15) & ~(size_t)15; };
            const size_t o_nraw = a16(sizeof(int) * (K + 3)),  /* off[K+1..K+2]: the pass's count of b words with set bits */ o_lit = o_nraw + a16(sizeof(int) * K), o_gap = o_lit + a16(sizeof(int) * K);
            const size_t o_fs = o_gap + a16(sizeof(double) * K), o_lines = o_fs + a16(sizeof(FrameScalars) * K);
            const size_t total = o_lines + sizeof(int) * 4 * K * (size_t)MAX_LINES;
            TRY(dev_alloc(s.d_res, total));
            TRY(host_alloc(s.h_res, total + 64));
            s.h_seq = (unsigned*)(s.h_res + total);  // behind everything a result copy ever touches
            *s.h_seq = 0;
            s.res_lines_off = o_lines;
            s.d_off = (int*)s.d_res;                 s.h_off = (int*)s.h_res;
            s.d_nraw = (int*)(s.d_res + o_nraw);     s.h_nraw = (int*)(s.h_res + o_nraw);
            s.d_litcopy = (int*)(s.d_res + o_lit);   s.h_litcount = (int*)(s.h_res + o_lit);
            s.d_gap = (double*)(s.d_res + o_gap);    s.h_gap = (double*)(s.h_res + o_gap);
            s.d_fscal = (FrameScalars*)(s.d_res + o_fs);  s.h_fscal = (FrameScalars*)(s.h_res + o_fs);
            s.d_packed = (int*)(s.d_res + o_lines);  s.h_lines = (int*)(s.h_res + o_lines);
        }
        HIPCHK(hipEventCreateWithFlags(&s.ev_pix, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
    }
#undef TRY
    HIPCHK(hipFuncSetAttribute((const void*)k_hough_ppht, hipFuncAttributeMaxDynamicSharedMemorySize,
                               (HOUGH_MAX_LIST + 61440 / 2 + 64) * (int)sizeof(unsigned)));
    // (the <true> instantiations carry the phase clocks of MDH_HOUGH_DEBUG and run only then)
    HIPCHK(hipFuncSetAttribute((const void*)k_hough3<false>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_large));
    HIPCHK(hipFuncSetAttribute((const void*)k_hough3_hybrid<false>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_large));
    HIPCHK(hipFuncSetAttribute((const void*)k_hough_single<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    HIPCHK(hipFuncSetAttribute((const void*)k_hough3<true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_large));
    HIPCHK(hipFuncSetAttribute((const void*)k_hough3_hybrid<true>, hipFuncAttributeMaxDynamicSharedMemorySize, c->lds_large));
    HIPCHK(hipFuncSetAttribute((const void*)k_hough_single<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

    // trig table exactly as OpenCV builds it: (float)(cos((double)n * theta) * irho), theta = (float)(pi/180), irho = 1
    float trig[2 * NUM_ANGLE];
    const float theta = (float)(3.14159265358979323846 / 180.0);
    for (int k = 0; k < NUM_ANGLE; ++k) {
        trig[2 * k] = (float)(cos((double)k * theta) * 1.0f);
        trig[2 * k + 1] = (float)(sin((double)k * theta) * 1.0f);
    }
    HIPCHK(hipMemcpy(c->d_trig, trig, sizeof(trig), hipMemcpyHostToDevice));
    ScalarState st{};
    static const double kA[3] = {2.0, 1.2, 0.9}, kB[3] = {4.4, 3.6, 3.0};
    static const int kInit[3] = {7, 5, 3};
    c->coef_a = kA[c->cfg.sensitivity];
    c->coef_b = kB[c->cfg.sensitivity];
    st.thr = c->cfg.adaptive_bi_thre ? kInit[c->cfg.sensitivity] : c->cfg.init_value;  // Detector.py:204-209
    st.thr_f = st.thr;
    HIPCHK(hipMemcpy(c->d_state, &st, sizeof(st), hipMemcpyHostToDevice));
    c->ema_m0 = 1 - (double)c->cfg.interval / 60;  // Detector.py:61
    c->ema_warm = c->n;                            // warmup_speed = n (Detector.py:62)
    c->ema_t = 0;
    c->hist.assign(c->n - 1, c->d_zero);
    return MDH_OK;
}

static bool env_on(const char* name) {
    const char* v = getenv(name);
    return v && v[0] == '1';
}

extern "C" mdh_ctx* mdh_create(int device, int H, int W, int n_window, const mdh_cfg* cfg) {
    if (!cfg) { set_err("mdh_create: null cfg"); return nullptr; }
    if (cfg->struct_size != sizeof(mdh_cfg)) {  // the only field read before this check is the first one
        set_err("mdh_create: cfg.struct_size is " + std::to_string(cfg->struct_size) + ", this library's mdh_cfg has " +
                std::to_string(sizeof(mdh_cfg)) + " bytes (ABI version " + std::to_string(MDH_ABI_VERSION) +
                "): the caller was built against another include/metdet_hip.h");
        return nullptr;
    }
    if (cfg->reserved != 0) { set_err("mdh_create: cfg.reserved must be 0"); return nullptr; }
    if (H <= 0 || W <= 0 || H > 32767 || W > 32767 || n_window < 1 || n_window > MDH_MAX_WINDOW) {
        set_err("mdh_create: bad geometry or window (need 1 <= n <= 255, 0 < H,W <= 32767)");
        return nullptr;
    }
    if (cfg->sensitivity < 0 || cfg->sensitivity > 2 || cfg->interval < 1 ||
        (cfg->detector != MDH_DET_M3 && cfg->detector != MDH_DET_CLASSIC)) {
        set_err("mdh_create: bad cfg (sensitivity / interval / detector)");
        return nullptr;
    }
    const int role = cfg->role;
    if (role != MDH_ROLE_FULL && role != MDH_ROLE_BAND && role != MDH_ROLE_HOUGH) {
        set_err("mdh_create: cfg.role must be MDH_ROLE_FULL, MDH_ROLE_BAND or MDH_ROLE_HOUGH");
        return nullptr;
    }
    if (role != MDH_ROLE_FULL && cfg->detector != MDH_DET_M3) {
        set_err("mdh_create: row-band tiling is built for the M3 detector only");
        return nullptr;
    }
    if (cfg->hough_mode != MDH_HOUGH_EXACT && cfg->hough_mode != MDH_HOUGH_ACCUMULATOR) {
        set_err("mdh_create: cfg.hough_mode must be MDH_HOUGH_EXACT or MDH_HOUGH_ACCUMULATOR");
        return nullptr;
    }
    const int* r = cfg->roi;
    // a band may hold no row of the noise ROI (empty roi: r[2] == r[0]); a Hough-only context never looks at it
    const bool roi_empty_ok = role == MDH_ROLE_BAND && r[2] == r[0] && r[3] >= r[1];
    if (role != MDH_ROLE_HOUGH && !roi_empty_ok &&
        (r[0] < 0 || r[1] < 0 || r[2] > H || r[3] > W || r[2] <= r[0] || r[3] <= r[1])) {
        set_err("mdh_create: std_roi outside the frame or empty");
        return nullptr;
    }
    if (cfg->init_value < 0 || cfg->init_value > 65535 || cfg->hough_threshold < 1 || cfg->hough_min_len < 0 ||
        cfg->hough_max_gap < 0) {
        set_err("mdh_create: init_value must be 0..65535, hough threshold >= 1, min_len / max_gap >= 0");
        return nullptr;
    }
    if (!(cfg->mask_area > 0)) {
        set_err("mdh_create: mask_area must be > 0");
        return nullptr;
    }
    // The context uses up to ten streams that should progress independently; the HIP runtime multiplexes streams onto
    // GPU_MAX_HW_QUEUES hardware queues (default 4), and two streams on one queue serialise behind each other's event
    // waits.  Correctness never depends on it (every dependency is an explicit event), only overlap does.  The variable is
    // read when the runtime initialises and belongs to the HOST process: the library does not touch the environment; hosts
    // that want the overlap export GPU_MAX_HW_QUEUES=12 before their first HIP call (bench.py does), and
    // mdh_hw_queues_hint() tells them what the library saw.
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_err("mdh_create: no HIP device visible -- libmetdet_hip needs an AMD GPU (there is no CPU fallback)");
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        set_err("mdh_create: device index out of range");
        return nullptr;
    }
    mdh_ctx* c = new mdh_ctx();
    c->device = device;
    c->H = H;
    c->W = W;
    c->WW = (W + 63) / 64;
    c->n = n_window;
    c->Kmax = cfg->max_batch > 0 ? cfg->max_batch : 64;
    if (const char* v = getenv("MDH_SLOTS")) c->nslot = std::max(2, std::min(NSLOT_MAX, atoi(v)));
    // the frame / binary rings must hold the history plus every pass in flight
    c->R = c->n - 1 + c->nslot * c->Kmax;
    if (c->R < c->n) c->R = c->n;
    c->HW = (size_t)H * W;
    c->words = (size_t)H * c->WW;
    c->cfg = *cfg;
    c->role = role;
    c->numrho = (W + H) * 2 + 1;
    c->force_generic = env_on("MDH_FORCE_GENERIC");
    c->force_bytes = env_on("MDH_FORCE_BYTES");
    c->force_ppl8 = env_on("MDH_FORCE_PPL8");
    c->hough_serial = env_on("MDH_HOUGH_SERIAL");
    c->hough_global_acc = env_on("MDH_HOUGH_GLOBAL_ACC");
    c->hough_one_wave = env_on("MDH_HOUGH_ONE_WAVE") || c->hough_serial;
    // Hough buffer / stream sets: one is enough while a pass's Hough stage is much shorter than its window kernel (1080p: +7 %
    // over two sets, fewer streams for the command processor to juggle); at 8K the hybrid frames make the stage as long as the
    // pixel stages and a second set lets two passes' Hough stages overlap (+15 %).
    c->nhset = (size_t)H * W > (size_t)12 << 20 ? 2 : 1;
    if (const char* v = getenv("MDH_HOUGH_SETS")) c->nhset = std::max(1, std::min(NHSET, atoi(v)));
    {   // one stream per pass: see mdh_ctx::one_stream
        const double max_mb = getenv("MDH_ONE_STREAM_MAX_MB") ? atof(getenv("MDH_ONE_STREAM_MAX_MB")) : 0.0;  // (off by default: measured 0.92 M against 1.02 M frames/s at 250-frame passes, +2 % at 1000: gpurun_out/r06_13, r06_16)
        c->one_stream = role == MDH_ROLE_FULL && cfg->detector == MDH_DET_M3 && (double)c->Kmax * H * W <= max_mb * 1e6 && c->Kmax > 8;
        if (c->one_stream) c->nhset = NHSET;
    }
    if (const char* v = getenv("MDH_HOUGH_BANDS")) c->hough_bands = atoi(v) != 0;
    if (const char* v = getenv("MDH_HOUGH_LDS_SMALL")) c->lds_small = std::max(16 * 1024, std::min(160 * 1024, atoi(v)));
    if (const char* v = getenv("MDH_HOUGH_LDS_LARGE")) c->lds_large = std::max(c->lds_small, std::min(160 * 1024, atoi(v)));
    // (frames of up to a megapixel only: at 960x540 the Hough stage is the largest item next to the window kernel and four
    // short-streak frames per CU are worth 2-6 % at the pipeline's pace; at 1080p and 4K the extra launch buys nothing or costs 1-2 %)
    if ((size_t)H * W > (size_t)1 << 20) c->lds_tiny = 0;
    if (const char* v = getenv("MDH_HOUGH_LDS_TINY")) c->lds_tiny = std::max(0, std::min(c->lds_small, atoi(v)));
    if (c->lds_tiny >= c->lds_small) c->lds_tiny = 0;
    if (role != MDH_ROLE_FULL) c->nmain = 1;  // (see post_stream)
    if (const char* v = getenv("MDH_MAIN_STREAMS")) c->nmain = std::max(1, std::min(2, atoi(v)));
    if (const char* v = getenv("MDH_CLASSIC_SPARSE")) c->classic_sparse_on = atoi(v) != 0;
    if (const char* v = getenv("MDH_CLASSIC_FAST4")) c->classic_fast4 = atoi(v) != 0;
    if (const char* v = getenv("MDH_LEAN_DIRECT")) c->lean_direct_on = atoi(v) != 0;
    if (const char* v = getenv("MDH_LEAN_POLL")) c->lean_poll = atoi(v) != 0;
    if (const char* v = getenv("MDH_LEAN_DUE")) c->lean_due_on = atoi(v) != 0;
    if (const char* v = getenv("MDH_LEAN_MAIL")) c->mail_on = atoi(v) != 0;
    if (const char* v = getenv("MDH_LEAN_MAIL_GAP")) c->mail_gap_us = atof(v);
    if (const char* v = getenv("MDH_AUTO_REGISTER")) c->auto_register_on = atoi(v) != 0;
    if (const char* v = getenv("MDH_DENSE")) c->dense_cfg = atoi(v) != 0;
    if (const char* v = getenv("MDH_DENSE_LEAVE")) c->dense_leave = atof(v);
    c->dense_now = c->dense_cfg == 1;

    if (create_impl(c) != MDH_OK) {
        std::string keep = g_err;
        free_all(c);
        set_err(keep);
        return nullptr;
    }
    return c;
}

extern "C" void mdh_destroy(mdh_ctx* c) { free_all(c); }

// ------------------------------------------------------------------------------------------------ profiling
// Timing events are taken from a per-context pool (creating and destroying a pair per kernel per pass cost more host
// time than the launches themselves).
static inline hipEvent_t pool_event(mdh_ctx* c) {
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}
static inline void span_begin(mdh_ctx* c, int kind, hipStream_t st) {
    if (!c->prof) return;
    ProfSpan s;
    s.kind = kind;
    s.a = pool_event(c);
    s.b = pool_event(c);
    hipEventRecord(s.a, st);
    c->cur_spans->push_back(s);
}
static inline void span_end(mdh_ctx* c, hipStream_t st) {
    if (!c->prof) return;
    hipEventRecord(c->cur_spans->back().b, st);
}
static void spans_collect(mdh_ctx* c, Slot& sl) {
    for (auto& s : sl.spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            c->prof_ms[s.kind] += ms;
            c->prof_n[s.kind] += 1;
        }
        if (g_host_log && c->ev_base) {  // GPU-side timeline without a profiler attached
            float t_a = 0, t_b = 0;
            hipEventElapsedTime(&t_a, c->ev_base, s.a);
            hipEventElapsedTime(&t_b, c->ev_base, s.b);
            fprintf(stderr, "[mdh gpu] %-16s start %.1f us  end %.1f us\n", kKernelNames[s.kind], t_a * 1e3, t_b * 1e3);
        }
        c->ev_pool.push_back(s.a);
        c->ev_pool.push_back(s.b);
    }
    sl.spans.clear();
}
extern "C" int mdh_profile_enable(mdh_ctx* c, int on) {
    if (!c) return MDH_E_ARG;
    c->prof_every = on > 0 ? on : 0;
    c->prof = on != 0;
    if (c->prof && g_host_log && !c->ev_base) {
        hipEventCreate(&c->ev_base);
        hipEventRecord(c->ev_base, c->stream);
        hipEventSynchronize(c->ev_base);
        host_log("gpu time base", 0);
    }
    return MDH_OK;
}
extern "C" int mdh_profile_read(mdh_ctx* c, double* ms, int64_t* launches, int reset) {
    if (!c) return MDH_E_ARG;
    for (int k = 0; k < MDH_NUM_KERNELS; ++k) {
        if (ms) ms[k] = c->prof_ms[k];
        if (launches) launches[k] = c->prof_n[k];
        if (reset) {
            c->prof_ms[k] = 0;
            c->prof_n[k] = 0;
        }
    }
    return MDH_OK;
}

// ------------------------------------------------------------------------------------------------ passes
// Post stream of the pass with sequence number `seq`.  Full contexts alternate between two (the post stages of pass p + 1 do not
// queue behind the Hough hand-over of pass p); band / Hough contexts of a tiled stream use ONE: a tiling runs 2 N contexts side by
// side, and every extra active stream is one more client of the command processor's few hardware queues (8 bands + 8 Hough
// contexts on one device: 104 active streams, 13 k frames/s).
// MDH_SKIP_STAGES (timing experiments only -- results are WRONG): bit 0 noise sums, bit 1 morphology, bit 2 dynamic mask, bit 3 Hough
static inline int skip_stages() {
    static const int v = getenv("MDH_SKIP_STAGES") ? atoi(getenv("MDH_SKIP_STAGES")) : 0;
    return v;
}
static inline hipStream_t post_stream(const mdh_ctx* c, int64_t seq) { return c->pstream[c->role == MDH_ROLE_FULL ? (seq & 1) : 0]; }

static PassGeom geom(const mdh_ctx* c, int K, int64_t t_base) {
    PassGeom g;
    g.H = c->H; g.W = c->W; g.WW = c->WW; g.n = c->n; g.K = K; g.Kcap = c->Kmax; g.t_base = t_base; g.R = c->R;
    return g;
}

// Window lengths above 16 have wide-load instantiations with 8 pixels per lane (round 6: the reference's default `window_sec: 1`
// is n = 25 / 30, Detector.py:200).  (MDH_WIDE8_MAXN=16 sends them back to round 1's byte-load kernel for A/B)
static bool long_window_wide(int n) {
    static const int wide8_max_n = getenv("MDH_WIDE8_MAXN") ? atoi(getenv("MDH_WIDE8_MAXN")) : 32;
    return n > 16 && n <= std::min(32, wide8_max_n);
}
// The window kernels' translation units (metdet_window.hip, one object per part of the window lengths): the part that holds the
// instantiation answers.
static int win_launch(const WindowLaunch& w) {
    return mdh_win_launch_p0(w) || mdh_win_launch_p1(w) || mdh_win_launch_p2(w) || mdh_win_launch_p3(w) || mdh_win_launch_p4(w) || mdh_win_launch_p5(w);
}
static int win_mail(const MailLaunch& m) {
    return mdh_win_mail_p0(m) || mdh_win_mail_p1(m) || mdh_win_mail_p2(m) || mdh_win_mail_p3(m) || mdh_win_mail_p4(m) || mdh_win_mail_p5(m);
}
// waves per SIMD an instantiation runs at (= resident 256-thread workgroups per CU): asked of the runtime once per (n, pixels per lane)
// (round 6: the pass tables moved several instantiations to another occupancy step; up to round 5 a table of register counts)
static int win_occupancy(int n, int ppl) {
    static int cache[33][2];
    if (n < 2 || n > 32) return 2;
    int& e = cache[n][ppl == 16];
    if (e == 0) {
        int v = 0;
        for (int p = 0; p < MDH_WIN_PARTS && !v; ++p) {
            switch (p) {
                case 0: v = mdh_win_occupancy_p0(n, ppl); break;
                case 1: v = mdh_win_occupancy_p1(n, ppl); break;
                case 2: v = mdh_win_occupancy_p2(n, ppl); break;
                case 3: v = mdh_win_occupancy_p3(n, ppl); break;
                case 4: v = mdh_win_occupancy_p4(n, ppl); break;
                default: v = mdh_win_occupancy_p5(n, ppl); break;
            }
        }
        e = v > 0 ? v : 2;
    }
    return e;
}
// Alignment the frames of the pass being queued allow the window kernel's loads: wide loads need contiguous 64-byte words
// (W % 64 == 0) and aligned frame bases.  >= 8 with 2 <= n <= 16: a wide-load variant runs (the only ones with a dense output).
static unsigned pass_alignment(const mdh_ctx* c, int K) {
    unsigned align = 16;
    if (c->W % 64 != 0 || c->force_bytes || c->force_generic) align = 1;
    for (int e = 0; e < c->n - 1 + K && align > 1; ++e)
        while (align > 1 && ((uintptr_t)c->h_ptrtab[e] & (align - 1))) align >>= 1;
    return align;
}
static bool wide_kernel_applies(const mdh_ctx* c, int K) {  // (the variants with a dense output: 16 pixels per lane up to n = 16, 8 per lane for n = 17 .. 32)
    static const int wide16_max_n = getenv("MDH_WIDE16_MAXN") ? atoi(getenv("MDH_WIDE16_MAXN")) : 16;
    const unsigned align = pass_alignment(c, K);
    if (c->n > 16) return align >= 8 && long_window_wide(c->n);
    return align >= 16 && c->n >= 2 && c->n <= std::min(16, wide16_max_n) && !c->force_ppl8;
}

// Momentum of the next EMA update (host mirror of utils.py:360-368; advances the schedule).
static double ema_next_momentum(mdh_ctx* c) {
    double m = c->ema_m0;
    if (c->ema_warm != 0) {
        const double ramp = (double)c->ema_t * (1 - c->ema_m0) * c->ema_warm;
        if (ramp < 1) {
            const double u = 1 - ramp;
            m = c->ema_m0 * (1 - u * u);
        } else {
            c->ema_warm = 0;
        }
    }
    c->ema_t += 1;
    return m;
}

// Stage K frames (absolute timers t_base+1 .. t_base+K): build the pointer table [hist (n-1) | pass frames (K)].
static int stage_frames(mdh_ctx* c, Slot& sl, const uint8_t* frames, int K, int on_device) {
    const int64_t t_base = sl.t_base;
    // first pass from host memory: the frame ring is allocated here, once (25 GB for 1080p passes of 2000 frames: not paid by callers
    // whose frames are resident in HBM).  Like every hipMalloc it synchronises the device -- a one-time stall of whatever other
    // contexts have in flight (the other one-time cost, the visiting-order tables of mdh_create, is built on a private stream)
    if (!on_device && !c->d_ring) HIPCHK(hipMalloc((void**)&c->d_ring, (size_t)c->R * c->HW));
    c->h_ptrtab = sl.h_ptrtab;
    for (int e = 0; e < c->n - 1; ++e) c->h_ptrtab[e] = c->hist[e];
    for (int i = 0; i < K; ++i) {
        const int64_t rs = (t_base + i) % c->R;
        c->h_ptrtab[c->n - 1 + i] = on_device ? frames + (size_t)i * c->HW : c->d_ring + (size_t)rs * c->HW;
    }
    for (int e = 0; e < WIN_PAD; ++e) c->h_ptrtab[c->n - 1 + K + e] = c->h_ptrtab[c->n - 2 + K];  // padding: the pass's last frame again
    if (sl.dense && !wide_kernel_applies(c, K)) sl.dense = false;  // (decided before k_pass_prepare: the sparse format wants its block counters zeroed)
    if (c->lean_direct) {
        // no copy commands at all: the table entry of the frame is the caller's (pinned) buffer, the table itself stays in
        // the slot's pinned header, and no sigma is due (checked by the caller), so there are no jobs
        c->h_ptrtab[c->n - 1] = c->lean_direct;
        c->d_ptrtab = sl.h_hdr_dev;
        sl.nj = 0;
        if (c->lean_due) {  // the frame's sigma is due: its one noise job sits in the pinned header as well
            NoiseJob& j = sl.h_jobs[0];
            j.frame = 0;
            j.k_used = c->n;
            j.m_cur = ema_next_momentum(c);
            j.one_minus = 1 - j.m_cur;
            sl.nj = 1;
            c->d_jobs = (NoiseJob*)((uint8_t*)sl.h_hdr_dev + sl.hdr_jobs_off);
        }
        return MDH_OK;
    }
    if (!on_device) {
        // contiguous host frames -> ring slots (t_base + i) % R : at most two runs
        int i = 0;
        while (i < K) {
            const int64_t rs = (t_base + i) % c->R;
            const int run = (int)std::min<int64_t>(K - i, c->R - rs);
            HIPCHK(hipMemcpyAsync(c->d_ring + (size_t)rs * c->HW, frames + (size_t)i * c->HW, (size_t)run * c->HW,
                                  hipMemcpyHostToDevice, c->nstream));
            i += run;
        }
        if (c->lean) HIPCHK(hipEventRecord(c->ev_copy, c->nstream));  // the caller's buffer has been read when this fires
    }
    // which frames are due for a noise sigma, and with which EMA momentum (host mirror of Detector.py:81-91, utils.py:360-368)
    const int n = c->n;
    const int64_t std_interval = (int64_t)c->cfg.interval * n;  // Detector.py:63
    int nj = 0;
    for (int i = 0; i < K; ++i) {
        const int64_t t = t_base + 1 + i;
        int k_used = 0;
        if (t > n && t % std_interval == 0) k_used = n;      // Detector.py:82
        else if (t > 1 && t <= n) k_used = (int)t;           // Detector.py:87
        if (!k_used) continue;
        const double m = ema_next_momentum(c);
        NoiseJob& j = sl.h_jobs[nj++];
        j.frame = i;
        j.k_used = k_used;
        j.m_cur = m;
        j.one_minus = 1 - m;
    }
    sl.nj = nj;
    // pointer table and jobs in one copy (lean mode: on a side stream, next to the frame's 2 MB instead of behind them)
    if (c->lean) {
        HIPCHK(hipMemcpyAsync(sl.d_hdr, sl.h_hdr, sl.hdr_jobs_off + sizeof(NoiseJob) * nj, hipMemcpyHostToDevice, c->cstream));
        HIPCHK(hipEventRecord(sl.ev_k1, c->cstream));
        HIPCHK(hipStreamWaitEvent(c->nstream, sl.ev_k1, 0));
        return MDH_OK;
    }
    HIPCHK(hipMemcpyAsync(sl.d_hdr, sl.h_hdr, sl.hdr_jobs_off + sizeof(NoiseJob) * nj, hipMemcpyHostToDevice, c->nstream));
    return MDH_OK;
}

// a1+a4+a6: time-blocked register-window kernel for 2 <= n <= 32, generic n-reads kernel otherwise.
template <int N>
static void launch_tb(mdh_ctx* c, const PassGeom& g) {
    const int words = (int)c->words;
    const int waves = (words + 3) / 4;
    int nchunks = (4096 + waves - 1) / waves;
    nchunks = std::max(1, std::min(nchunks, g.K / (2 * N)));
    const int Tc = (g.K + nchunks - 1) / nchunks;
    nchunks = (g.K + Tc - 1) / Tc;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stack_threshold_tb<N>), dim3((waves + 3) / 4, nchunks), dim3(256), 0, c->stream,
                       c->d_ptrtab, c->d_thr, c->d_bbits, c->d_bandflag, g, Tc);
}

// MDH_LEAN_MAIL_CLEAN=0: k_hough_single keeps cleaning up behind its own results (A/B)
static bool mail_clean_on() {
    static const bool on = !(getenv("MDH_LEAN_MAIL_CLEAN") && atoi(getenv("MDH_LEAN_MAIL_CLEAN")) == 0);
    return on;
}
// The window kernel waiting on the mailbox (if any) is told to leave; it is gone within a poll period, whatever is queued next on
// the staging stream follows it in order.
static void mail_cancel(mdh_ctx* c) {
    if (!c->mail_armed) return;
    __atomic_store_n(&c->h_mail->line[0][15], ~c->mail_seq, __ATOMIC_RELEASE);
    c->mail_armed = false;
    c->mail_misses += 1;
}
// Queue the NEXT frame's window kernel (end of mdh_update): n = 2..16 with 16 pixels per lane, n = 17..32 with 8, one chunk -- what
// launch_wide<N, 16 | 8> launches for a direct frame.
static constexpr int mail_ppl(int n) { return n > 16 ? 8 : 16; }
static void mail_prelaunch(mdh_ctx* c, const MailClean& cl) {
    const int words = (int)c->words;
    const int ppl = mail_ppl(c->n);
    const unsigned grid = (unsigned)(((words + ppl - 1) / ppl + 3) / 4);
    c->mail_seq += 1;
    if (c->mail_seq == 0 || c->mail_seq == ~0u) c->mail_seq = 1;
    unsigned* go = c->d_read_count + 32;
    if (c->n > 16 && !long_window_wide(c->n)) return;
    if (!win_mail(MailLaunch{c->n, ppl, grid, c->nstream, c->h_mail, c->mail_seq, go, cl})) return;
    if (hipGetLastError() != hipSuccess) return;  // (not queued: the next frame is launched the usual way)
    c->mail_armed = true;
    c->mail_n = c->n;
    c->mail_grid = grid;
}
// One launch of a wide-load window kernel, N = g.n frames per window, PPL = 16 or 8 pixels per lane (the instantiations live in
// the window translation units: metdet_window.hip)
static void launch_wide(mdh_ctx* c, const PassGeom& g, const LeanArgs& lean, const int PPL) {
    const int N = g.n;
    const int words = (int)c->words;
    const int waves = (words + PPL - 1) / PPL;
    static const int target_waves = getenv("MDH_WIN_WAVES") ? atoi(getenv("MDH_WIN_WAVES")) : 4096;  // experiment switch
    static const int wpb = getenv("MDH_WIN_BLOCK") ? std::max(1, std::min(4, atoi(getenv("MDH_WIN_BLOCK")) / 64)) : 4;  // waves per workgroup
    // Short passes (under ~3 GB of frames: their window kernel lasts a few hundred microseconds) are cut into four times as many time
    // chunks: the workgroups are shorter-lived, and the small kernels of the neighbouring passes -- which wait for a workgroup slot
    // behind them -- get one sooner (1000-frame 1080p passes 1.94 -> 2.10 M frames/s, 250-frame ones +3 %: gpurun_out/r06_24; a
    // 4000-frame pass gains nothing and pays the extra window histories).
    const bool short_pass = (double)g.K * (double)c->HW <= 3.0e9 && !getenv("MDH_WIN_WAVES");
    int nchunks = ((short_pass ? 4 * target_waves : target_waves) + waves - 1) / waves;
    const int cap = std::max(1, g.K / (2 * N));
    nchunks = std::max(1, std::min(nchunks, cap));
    {   // Time chunks so that the grid is close to a whole number of ROUNDS of resident workgroups (256 CUs x 3 workgroups of four
        // waves at <= 168 VGPRs, x 2 above): a band of an 8K frame (548 rows: 1028 workgroups) in one chunk is 1.34 rounds -- the
        // launch lasts two, a third of the machine idle in the second -- in two chunks 2.68 of 3.  Whole frames at the bench's
        // sizes are unchanged (1080p: 3 chunks = 1.98 rounds; 4K / n = 16: 3.95; 8K: 10.5).
        // Round 4: on by default (MDH_WIN_BALANCE=0 for A/B).  Round 3 measured it slower at 960x540 (4.03 -> 3.76 M frames/s) when the
        // pass was bound by its other stages; with those out of the way it is the launch's own tail that shows: 960x540 in 9 chunks
        // is 1143 workgroups on 768 resident slots, 1.49 rounds (the isolated launch 0.47 -> 0.53-0.56 of the peak with 12-16 chunks,
        // +2 % at the pipeline's pace, where the next pass's launch fills the tail anyway).
        static const bool balance = !(getenv("MDH_WIN_BALANCE") && atoi(getenv("MDH_WIN_BALANCE")) == 0);
        // waves per SIMD the instantiation runs at (= resident 256-thread workgroups per CU): asked of the runtime once (round 6: the
        // pass tables moved several instantiations to another occupancy step; up to round 5 a table of register counts)
        const int occ = win_occupancy(N, PPL);
        const double resident = 256.0 * occ * 4 / wpb;
        const double wgs = (double)((waves + wpb - 1) / wpb);
        auto eff = [&](int nc) { const double r = wgs * nc / resident; return r / std::ceil(r); };
        if (balance) {
            int best = nchunks;
            for (int nc = nchunks + 1; nc <= std::min(cap, 2 * nchunks + 2); ++nc)
                if (eff(nc) > eff(best) + 0.04) best = nc;
            nchunks = best;
        }
    }
    int Tc = (g.K + nchunks - 1) / nchunks;
#if WIDE_TABLES
    if (!lean.copy_dst) Tc = (Tc + N - 1) / N * N;  // every time chunk but the last: whole window periods (the kernel tests nothing per frame, see WinRec)
#endif
    nchunks = (g.K + Tc - 1) / Tc;
    if (lean.copy_dst) {  // per-frame protocol: its own instantiation, the batched one pays nothing for it
        if (c->mail_armed && PPL == mail_ppl(N) && c->mail_n == N && nchunks == 1 && c->mail_grid == (unsigned)((waves + 3) / 4) && wpb == 4 &&
            c->stream == c->nstream && __atomic_load_n(&c->h_mail->state, __ATOMIC_ACQUIRE) != ~c->mail_seq) {
            // the kernel is waiting (or about to be resident): hand it the frame
            unsigned pay[MAIL_PAYLOAD] = {0};
            MailArgs a{};
            // the kernel reads the pass's pointer table from the payload's copy in device memory, not across the bus
            a.ptrtab = (const uint8_t* const*)(c->d_read_count + 64 + MAIL_ARG_DW);
            a.thr_tab = c->d_thr; a.bbits = c->d_bbits; a.blockcnt = c->d_bandflag; a.g = g; a.Tc = Tc; a.lean = lean;
            memcpy(pay, &a, sizeof(a));
            memcpy(pay + MAIL_ARG_DW, c->h_ptrtab, sizeof(uint8_t*) * (size_t)(N - 1 + 1));
            const int nch = (MAIL_ARG_DW + 2 * N + 14) / 15;
            for (int ch = 0; ch < nch; ++ch) memcpy((void*)&c->h_mail->line[ch][0], pay + 15 * ch, 60);
            __atomic_thread_fence(__ATOMIC_RELEASE);
            for (int ch = nch - 1; ch >= 0; --ch) __atomic_store_n(&c->h_mail->line[ch][15], mail_line_tag(pay + 15 * ch, c->mail_seq), __ATOMIC_RELEASE);
            c->mail_armed = false;
            // Taken or given up?  The kernel decides once (one lane) and says so: the frame's other launches are only queued
            // behind a window kernel that IS working on the frame.  (The wait is a bus round trip while the frame's 2 MB start
            // to cross it; a kernel that has not become resident yet answers when it has; one that expired a moment ago has
            // answered already.)
            unsigned st;
            const double t_spin = now_us();
            for (int spin = 0;; ++spin) {
                st = __atomic_load_n(&c->h_mail->state, __ATOMIC_ACQUIRE);
                if (st == c->mail_seq || st == ~c->mail_seq) break;
                __builtin_ia32_pause();
                if ((spin & 4095) == 4095 && now_us() - t_spin > 2e6) {  // (a kernel that neither takes its frame nor expires: the device is gone)
                    set_err("the window kernel waiting for this frame does not answer");
                    c->poisoned = true;
                    __atomic_store_n(&c->h_mail->line[0][15], ~c->mail_seq, __ATOMIC_RELEASE);  // a kernel that turns up late gives up
                    st = ~c->mail_seq;
                    break;
                }
            }
            c->mail_used = st == c->mail_seq;
            if (c->mail_used) c->mail_hits += 1; else c->mail_misses += 1;
        } else {
            mail_cancel(c);
        }
        if (!c->mail_used)
            c->win_missing |= !win_launch(WindowLaunch{N, PPL, WIN_LEAN, dim3((waves + 3) / 4, nchunks), dim3(256), c->tag_lds, c->stream, c->d_ptrtab, c->d_thr, c->d_bbits,
                                    c->d_bandflag, g, Tc, lean, nullptr, nullptr});
    }
    else if (wide_has_dense(N, PPL) && c->win_dense)  // dense output format (its own instantiation as well)
        c->win_missing |= !win_launch(WindowLaunch{N, PPL, WIN_DENSE, dim3((waves + wpb - 1) / wpb, nchunks), dim3(64 * wpb), c->tag_lds, c->stream, c->d_ptrtab, c->d_thr,
                                (ActWord*)c->win_plain, dense_candidates_inline() ? c->d_bandflag : (unsigned*)nullptr, g, Tc, lean, c->win_nzwords, c->d_wrec});
    else
        c->win_missing |= !win_launch(WindowLaunch{N, PPL, WIN_SPARSE, dim3((waves + wpb - 1) / wpb, nchunks), dim3(64 * wpb), c->tag_lds, c->stream, c->d_ptrtab, c->d_thr,
                                c->d_bbits, getenv("MDH_ALL_DENSE") ? (unsigned*)nullptr : c->d_bandflag, g, Tc, lean, c->win_nzwords, c->d_wrec});
    c->win_dense_used = wide_has_dense(N, PPL) && c->win_dense && !lean.copy_dst;
}

// copy_dst: see k_stack_threshold_wide (per-frame protocol; the caller has made sure a wide-load variant applies)
static void launch_stack_threshold(mdh_ctx* c, const PassGeom& g, const LeanArgs& copy_dst = LeanArgs{}) {
    const int words = (int)c->words;
    if (c->force_generic) {
        hipLaunchKernelGGL(k_stack_threshold_generic, dim3((words + 3) / 4, g.K), dim3(256), 0, c->stream, c->d_ptrtab, c->d_thr,
                           c->d_bbits, c->d_bandflag, g);
        return;
    }
    const unsigned align = pass_alignment(c, g.K);
    // 16 pixels per lane up to n = 16 (252 VGPRs at n = 16: still two waves per SIMD; 4K / n = 16: 1.35 -> 0.92 ms per 512 frames)
    static const int wide16_max_n = getenv("MDH_WIDE16_MAXN") ? atoi(getenv("MDH_WIDE16_MAXN")) : 16;  // A/B switch
    if (align >= 16 && g.n >= 2 && g.n <= std::min(16, wide16_max_n) && !c->force_ppl8) {
        launch_wide(c, g, copy_dst, 16);
        return;
    }
    // windows of 17 .. 32 frames (round 6; the reference's default `window_sec: 1` at 25 / 30 frames per second, Detector.py:200):
    // 8 pixels per lane -- the window of a 16-pixel lane does not fit two waves per SIMD (MDH_WIDE8_MAXN=16 for A/B: round 1's byte-load kernel)
    if (align >= 8 && g.n >= 2 && (g.n <= 16 || long_window_wide(g.n))) {
        launch_wide(c, g, copy_dst, 8);
        return;
    }
    switch (g.n) {
#define TB_CASE(N) case N: launch_tb<N>(c, g); break;
        TB_CASE(2) TB_CASE(3) TB_CASE(4) TB_CASE(5) TB_CASE(6) TB_CASE(7) TB_CASE(8) TB_CASE(9) TB_CASE(10) TB_CASE(11)
        TB_CASE(12) TB_CASE(13) TB_CASE(14) TB_CASE(15) TB_CASE(16) TB_CASE(17) TB_CASE(18) TB_CASE(19) TB_CASE(20)
        TB_CASE(21) TB_CASE(22) TB_CASE(23) TB_CASE(24) TB_CASE(25) TB_CASE(26) TB_CASE(27) TB_CASE(28) TB_CASE(29)
        TB_CASE(30) TB_CASE(31) TB_CASE(32)
#undef TB_CASE
        default:
            hipLaunchKernelGGL(k_stack_threshold_generic, dim3((words + 3) / 4, g.K), dim3(256), 0, c->stream, c->d_ptrtab,
                               c->d_thr, c->d_bbits, c->d_bandflag, g);
    }
}

// LineDetector.update for the K staged frames, first half: which frames are due, and their noise sums over cfg.roi.
// Returns the number of due frames in *nj_out.  (Row-band tiling all-reduces the sums between the two halves.)
static int noise_prepare(mdh_ctx* c, Slot& sl, int K, int* nj_out) {
    const int64_t t_base = sl.t_base;
    const int nj = sl.nj;  // built by stage_frames
    const int* r = c->cfg.roi;
    const int rh = r[2] - r[0], rw = r[3] - r[1];
    PassGeom g = geom(c, K, t_base);
    // One launch clears the counters the pixel stages accumulate into, the block bit counts, the noise sums and the dst
    // words this slot's previous pass left behind (its last reader, that pass's Hough stage, is done: the pass has been
    // collected).
    const size_t nb = (size_t)(c->H + MORPH_RB - 1) / MORPH_RB;
    const bool m3 = c->cfg.detector == MDH_DET_M3;
    if (!m3 && c->role == MDH_ROLE_FULL) {
        // ClassicDetector: the format of this pass (see mdh_ctx::classic_sparse_on); a slot whose last pass ran the dense kernels has
        // every word of its dst written: cleared in full before its first sparse pass (the sparse format only ever clears what it listed)
        static const bool classic_tb = !(getenv("MDH_CLASSIC_TB") && atoi(getenv("MDH_CLASSIC_TB")) == 0);
        sl.classic_sparse = c->classic_sparse_on && sl.d_cplanes != nullptr && classic_tb && c->W % 64 == 0 && !c->force_bytes && !c->force_generic &&
                            (c->n == 3 || c->n == 4) && c->classic_dense_passes <= 0;
        if (c->classic_dense_passes > 0) c->classic_dense_passes -= 1;
        if (sl.classic_sparse && sl.classic_dense_last) HIPCHK(hipMemsetAsync(sl.d_dst, 0, sizeof(u64) * (size_t)c->Kmax * c->words, c->nstream));
    }
    hipLaunchKernelGGL(k_pass_prepare, dim3(1024), dim3(256), 0, c->nstream, (unsigned*)sl.d_rowcount,
                       (unsigned)((size_t)c->Kmax * c->H + (6 + WORKCNT_STRIDE) * (size_t)c->Kmax + 1), sl.d_bandflag,
                       (unsigned)(nb * c->WW), (unsigned)c->Kmax,  // (only what the slot's previous pass touched)
                       (unsigned*)c->d_sums, (unsigned)(4 * nj), sl.d_dst, sl.d_nzlist,
                       ((m3 || (sl.d_cplanes != nullptr && !sl.classic_dense_last)) && sl.used) ? sl.d_nzcount : nullptr);
    if (nj) {
        bool aligned4 = (c->W % 4) == 0;
        for (int e = 0; e < c->n - 1 + K && aligned4; ++e) aligned4 = ((uintptr_t)c->h_ptrtab[e] & 3) == 0;
        span_begin(c, KK_NOISE_SUMS, c->nstream);
        if (rh <= 0 || rw <= 0) {
            // row-band tiling: this band holds no row of the noise ROI; its share of the sums is zero
        } else if (skip_stages() & 1) {
        } else if (aligned4) {
            const int xa = r[1] & ~3;
            const int nx4 = (r[1] + rw - xa + 3) >> 2;
            // (at most 48 workgroups per due frame; 8 or 16 at 960x540 -- 1000 due frames per pass -- shorten the launch alone, 0.235 -> 0.176 ms,
            // and change nothing at the pipeline's pace)
            hipLaunchKernelGGL(k_noise_sums<0>, dim3(std::min((rh * nx4 + 255) / 256, 48), nj), dim3(256), 0, c->nstream, c->d_ptrtab,
                               c->d_jobs, c->d_sums, g, r[0], r[1], rh, rw);
        } else {
            hipLaunchKernelGGL(k_noise_sums_bytes, dim3((rh * rw + 255) / 256, nj), dim3(256), 0, c->nstream, c->d_ptrtab,
                               c->d_jobs, c->d_sums, g, r[0], r[1], rh, rw);
        }
        span_end(c, c->nstream);
        if (!c->injected.empty()) {
            // sums the caller knows better (mdh_inject_sums) replace what the kernel made of an incomplete window
            int staged = 0;  // at most 64 entries exist, so at most 64 are staged per pass
            for (int j = 0; j < nj; ++j) {
                const int64_t t = t_base + 1 + sl.h_jobs[j].frame;
                for (size_t q = 0; q < c->injected.size(); ++q) {
                    if (c->injected[q].first != t) continue;
                    long long* stage = c->h_inject + ((size_t)(&sl - c->slot) * 64 + (size_t)(staged++ % 64)) * 2;
                    stage[0] = c->injected[q].second.first;
                    stage[1] = c->injected[q].second.second;
                    HIPCHK(hipMemcpyAsync(c->d_sums + 2 * j, stage, sizeof(long long) * 2, hipMemcpyHostToDevice, c->nstream));
                    c->injected.erase(c->injected.begin() + q);
                    break;
                }
            }
        }
    }
    HIPCHK(hipGetLastError());
    *nj_out = nj;
    return MDH_OK;
}

// Second half: the scalar chain (EMA, threshold) over the due frames from their sums (roi_px = pixels the sums cover),
// then everything the pixel stages of the pass wait for.
static int noise_finish(mdh_ctx* c, Slot& sl, int K, int nj, double roi_px) {
    span_begin(c, KK_NOISE_FINAL, c->nstream);
    hipLaunchKernelGGL(k_noise_finalize, dim3(1), dim3(NF_THREADS), c->tag_lds, c->nstream, c->d_jobs, nj, c->d_sums, c->d_state, c->d_jobres,
                       sl.d_fscal, c->d_thr, K, roi_px, c->cfg.adaptive_bi_thre, c->coef_a, c->coef_b, sl.d_nzcount,
                       (c->seek_at >= 0 && c->cfg.detector == MDH_DET_M3) ? (int)std::max<int64_t>(0, std::min<int64_t>(K, c->seek_at + c->n - 1 - sl.t_base)) : 0,
                       // (ClassicDetector's fast kernel reads its thresholds from the same records: len = 1, so the low half is T itself)
                       sl.d_wrec, c->cfg.detector == MDH_DET_M3 ? c->n : 1, (long long)sl.t_base);
    span_end(c, c->nstream);
    HIPCHK(hipGetLastError());
    EV_REC(c, sl.ev_noise, c->nstream);
    return MDH_OK;
}

static int run_noise(mdh_ctx* c, Slot& sl, int K) {
    int nj = 0, rc;
    if ((rc = noise_prepare(c, sl, K, &nj)) != MDH_OK) return rc;
    const int* r = c->cfg.roi;
    return noise_finish(c, sl, K, nj, (double)(r[2] - r[0]) * (r[3] - r[1]));
}

// Per-frame protocol, a frame whose sigma is due, read straight from the caller's pinned buffer (round 3: such a frame -- one in
// interval * n -- used to fall back to the general path: DMA copy with its 17 us hand-over, header copy, k_pass_prepare, noise chain).
// Two launches in front of the window kernel: the ROI sums (the new frame's ROI crosses PCIe here, 0.2 MB) and the scalar chain.
// The slot is clean (lean_direct_source checked it) and its sums are zero: k_hough_single's clean-up leaves them so.
static int run_noise_direct(mdh_ctx* c, Slot& sl) {
    const int* r = c->cfg.roi;
    const int rh = r[2] - r[0], rw = r[3] - r[1];
    PassGeom g = geom(c, 1, sl.t_base);
    const int xa = r[1] & ~3;
    const int nx4 = (r[1] + rw - xa + 3) >> 2;
    span_begin(c, KK_NOISE_SUMS, c->nstream);
    hipLaunchKernelGGL(k_noise_sums<0>, dim3(std::min((rh * nx4 + 255) / 256, 48), 1), dim3(256), 0, c->nstream, c->d_ptrtab, c->d_jobs, c->d_sums, g,
                       r[0], r[1], rh, rw);
    span_end(c, c->nstream);
    span_begin(c, KK_NOISE_FINAL, c->nstream);
    hipLaunchKernelGGL(k_noise_finalize, dim3(1), dim3(NF_THREADS), 0, c->nstream, c->d_jobs, 1, c->d_sums, c->d_state, c->d_jobres, sl.d_fscal, c->d_thr, 1,
                       (double)rh * rw, c->cfg.adaptive_bi_thre, c->coef_a, c->coef_b, (unsigned*)nullptr, 0, (WinRec*)nullptr, c->n, (long long)sl.t_base);
    span_end(c, c->nstream);
    HIPCHK(hipGetLastError());
    return MDH_OK;
}

// {M3,Classic}Detector.detect for the K staged frames, pixel stages: window kernel on the main stream, morphology and
// dynamic mask on the pass's post stream.  Leaves dst bit planes, row / frame lit counts and bounding boxes in the slot.
static int run_pixels(mdh_ctx* c, Slot& sl, int K) {
    const int64_t t_base = sl.t_base;
    PassGeom g = geom(c, K, t_base);
    const int words = (int)c->words;
    const size_t nb = (size_t)(c->H + MORPH_RB - 1) / MORPH_RB;
    // Main stream: nothing but the window kernel (the one stage that fills the machine), pass after pass.  Everything
    // downstream of it -- morphology, dynamic mask, Hough, result packing -- is short or latency-bound and runs on the
    // pass's post streams, overlapping the window kernel of the following pass(es).
    // (the post streams are NOT the Hough streams: morphology of pass p must not queue behind the Hough stage of pass p-2)
    hipStream_t post = c->lean ? c->nstream : (c->one_stream ? c->stream : post_stream(c, c->pass_seq));
    if (!c->lean_direct) EV_WAIT(c, c->stream, sl.ev_noise);  // thresholds, pointer table, history copies, cleared counters
    if (c->cfg.detector == MDH_DET_M3) {
        span_begin(c, KK_STACK_THR, c->stream);
        static const bool skip_window = getenv("MDH_SKIP_WINDOW") != nullptr;  // timing experiments only: results are wrong
        LeanArgs copy_dst{};
        c->win_dense = sl.dense && !c->lean;
        c->win_dense_used = false;
        c->win_plain = sl.d_bplain;
        c->win_nzwords = c->lean ? nullptr : (unsigned long long*)(sl.d_nzcount + 2);
        if (c->lean_direct) {  // the threshold is the detector's running one (or, sigma due, what the short chain just made), the ring copy is the kernel's job
            if (!c->lean_due) c->d_thr = &c->d_state->thr;
            c->read_seq += 1;
            if (c->read_seq == 0) c->read_seq = 1;
            copy_dst = LeanArgs{c->d_ring + (size_t)(t_base % c->R) * c->HW, sl.d_dst, sl.d_nzlist, sl.d_nzcount,
                                c->lean_poll ? c->d_read_count : nullptr, c->h_read_flag, c->read_seq};
        }
        if (!skip_window) launch_stack_threshold(c, g, copy_dst);  // (nothing stored = all zero: the words are tagged)
        span_end(c, c->stream);
        if (c->win_missing) {
            set_err("internal: no window-kernel instantiation for n = " + std::to_string(g.n) + " (metdet_window.hip parts out of step with launch_wide)");
            return MDH_E_STATE;
        }
        // the caller's buffer has been read when the kernel is done: announced by the kernel itself (polled word), or by an event
        if (c->lean_direct && !c->lean_poll) HIPCHK(hipEventRecord(c->ev_copy, c->stream));
    } else {
        span_begin(c, KK_CLASSIC, c->stream);
        // time-blocked wide-load kernel when the layout allows it (rows of whole 64-pixel words, 16-byte aligned frames, a ring of
        // 3 or 4 frames); MDH_CLASSIC_TB=0 keeps the one-lane-per-pixel kernel (A/B and parity testing)
        static const bool classic_tb = !(getenv("MDH_CLASSIC_TB") && atoi(getenv("MDH_CLASSIC_TB")) == 0);
        bool wide = classic_tb && c->W % 64 == 0 && !c->force_bytes && !c->force_generic && (g.n == 3 || g.n == 4) && K >= 1;
        for (int e = 0; e < c->n - 1 + K && wide; ++e) wide = ((uintptr_t)c->h_ptrtab[e] & 15) == 0;
        if (sl.classic_sparse && !wide) sl.classic_sparse = false;  // (a misaligned frame in this very pass: the dense kernels, which write every word of dst)
        sl.classic_dense_last = !sl.classic_sparse;
        if (wide) {
            const int waves = (words + 15) / 16;
            int nchunks = std::max(1, std::min((8192 + waves - 1) / waves, K / 8));
            const int Tc = (K + nchunks - 1) / nchunks;
            nchunks = (K + Tc - 1) / Tc;
            if (sl.classic_sparse) {
                if (g.n == 4 && c->classic_fast4) {
                    const int Tc4 = (Tc + 3) / 4 * 4;  // a time chunk is a whole number of four-frame bursts
                    hipLaunchKernelGGL(k_classic_diff4_sparse, dim3((waves + 3) / 4, (K + Tc4 - 1) / Tc4), dim3(256), 0, c->stream, c->d_ptrtab,
                                       (const WinRec*)sl.d_wrec, sl.d_cplanes, sl.d_bandflag, g, Tc4);
                } else if (g.n == 4)
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_classic_diff_tb<4, true>), dim3((waves + 3) / 4, nchunks), dim3(256), 0, c->stream, c->d_ptrtab,
                                       c->d_thr, sl.d_planes, g, Tc, sl.d_cplanes, sl.d_bandflag);
                else
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_classic_diff_tb<3, true>), dim3((waves + 3) / 4, nchunks), dim3(256), 0, c->stream, c->d_ptrtab,
                                       c->d_thr, sl.d_planes, g, Tc, sl.d_cplanes, sl.d_bandflag);
            } else
            if (g.n == 4)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_classic_diff_tb<4>), dim3((waves + 3) / 4, nchunks), dim3(256), 0, c->stream, c->d_ptrtab,
                                   c->d_thr, sl.d_planes, g, Tc, (ActWord*)nullptr, (unsigned*)nullptr);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_classic_diff_tb<3>), dim3((waves + 3) / 4, nchunks), dim3(256), 0, c->stream, c->d_ptrtab,
                                   c->d_thr, sl.d_planes, g, Tc, (ActWord*)nullptr, (unsigned*)nullptr);
        } else {
            hipLaunchKernelGGL(k_classic_diff, dim3((words + 3) / 4, K), dim3(256), 0, c->stream, c->d_ptrtab, c->d_thr, sl.d_planes, g);
        }
        span_end(c, c->stream);
    }
    EV_REC(c, sl.ev_k1, c->stream);
    EV_WAIT(c, post, sl.ev_k1);
    // (dst of this slot is all-zero again: k_pass_prepare ran before ev_noise, which the window kernel waited for)
    if (c->cfg.detector == MDH_DET_M3) {
        span_begin(c, KK_MORPH, post);
        if (c->win_dense_used) {
            c->dense_passes += 1;
            static const int plain_grid = getenv("MDH_PLAIN_GRID") ? std::max(64, atoi(getenv("MDH_PLAIN_GRID"))) : 8192;
            if (dense_candidates(c)) {
                // dense format, round 5: one streaming sweep over the plain words flags the blocks that hold a pair of set bits close
                // enough for the median to keep anything (0.03 % of the words on 8K noise: k_dense_candidates); their 3x3 block
                // neighbourhoods are listed and closed from the plain words (round 4: every word of every frame through
                // k_morph_close_plain's dependent row loads, 0.96 ms of a 4.5 ms pass).  Frames whose list covers half of their
                // blocks still go through that kernel, by a list of their own.
                const size_t bm_dwords = (nb * c->WW + 31) / 32;
                const int wpf = bm_dwords > 256 ? 4 : (bm_dwords > 64 ? 2 : 1);
                static const int cand_grid = getenv("MDH_CAND_GRID") ? std::max(64, atoi(getenv("MDH_CAND_GRID"))) : 4096;
                if (!dense_candidates_inline())
                    hipLaunchKernelGGL(k_dense_candidates, dim3(cand_grid), dim3(256), 0, post, sl.d_bplain, sl.d_bandflag, g);
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_morph_select_touched<false>), dim3((unsigned)((K + 4 / wpf - 1) / (4 / wpf))), dim3(256), 0, post,
                                   sl.d_bandflag, sl.d_worklist, sl.d_workcnt, g, wpf, dense_candidates_inline() ? 0 : 1);
                // (the list kernel is a chain of dependent steps per workgroup: enough one-wave workgroups per frame that a frame's
                // few thousand listed blocks are a few dozen steps each)
                const int lwpf = (int)std::max<long long>(wpf, std::min<long long>(32, 16384 / std::max(K, 1)));
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_morph_close_t<true>), dim3((unsigned)std::min(K * lwpf, 16384)), dim3(64), 0, post, (const ActWord*)sl.d_bplain,
                                   sl.d_worklist, sl.d_workcnt, c->d_act, sl.d_nzlist, sl.d_nzcount, g, sl.d_dense, lwpf);
                hipLaunchKernelGGL(k_morph_close_plain, dim3(plain_grid), dim3(64), 0, post, sl.d_bplain, (const unsigned*)sl.d_dense, c->d_act, sl.d_nzlist,
                                   sl.d_nzcount, g);
            } else {
                // (MDH_DENSE_CAND=0, A/B and parity: every frame through the 64-bit direct-mapped kernel, no work lists)
                hipLaunchKernelGGL(k_morph_close_plain, dim3(plain_grid), dim3(64), 0, post, sl.d_bplain, (const unsigned*)nullptr, c->d_act, sl.d_nzlist,
                                   sl.d_nzcount, g);
            }
        } else if (c->lean && K == 1) {
            // one frame: fourteen rows of a block on fourteen lanes (MDH_MORPH_T=0: the one-lane row pipeline of round 2, A/B and parity testing)
            static const bool transposed = !(getenv("MDH_MORPH_T") && atoi(getenv("MDH_MORPH_T")) == 0);
            if (transposed)
                hipLaunchKernelGGL(k_morph_close_frame_t, dim3((unsigned)((nb * c->WW + 63) / 64)), dim3(64), 0, post, sl.d_bbits, sl.d_bandflag,
                                   c->d_act, sl.d_nzlist, sl.d_nzcount, g);
            else
            hipLaunchKernelGGL(k_morph_close_frame, dim3((unsigned)((nb * c->WW + 63) / 64)), dim3(64), 0, post, sl.d_bbits, sl.d_bandflag, c->d_act,
                               sl.d_nzlist, sl.d_nzcount, g);
        } else if (skip_stages() & 2) {
        } else {
        static const bool all_dense = getenv("MDH_ALL_DENSE") != nullptr;  // experiment: no block counters, every frame direct-mapped
        if (all_dense) {
            hipLaunchKernelGGL(k_morph_close_dense, dim3(8192), dim3(64), 0, post, sl.d_bbits, (const unsigned*)nullptr, c->d_act, sl.d_nzlist, sl.d_nzcount, g);
        } else {
        // waves (select) / one-wave workgroups (close) per frame: by the size of a frame's bitmap (MDH_MORPH_WPF: 1, 2, 4; A/B)
        static const int wpf_env = getenv("MDH_MORPH_WPF") ? atoi(getenv("MDH_MORPH_WPF")) : 0;
        const size_t bm_dwords = (nb * c->WW + 31) / 32;
        const int wpf = (wpf_env == 1 || wpf_env == 2 || wpf_env == 4) ? wpf_env : (bm_dwords > 256 ? 4 : (bm_dwords > 64 ? 2 : 1));
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_morph_select_touched<true>), dim3((unsigned)((K + 4 / wpf - 1) / (4 / wpf))), dim3(256), 0, post, sl.d_bandflag, sl.d_worklist, sl.d_workcnt, g, wpf, 1);
        // frames with at least half of their blocks listed (dense binaries: 8K at m3det_high) go to the direct-mapped kernel, whose
        // waves leave after one load when the pass has none (MDH_MORPH_DENSE=0: everything through the list, A/B and parity testing)
        static const bool dense = !(getenv("MDH_MORPH_DENSE") && atoi(getenv("MDH_MORPH_DENSE")) == 0);
        // a block's fourteen rows on fourteen lanes, one wave per frame (MDH_MORPH_BT=0: one lane per listed block, A/B and parity)
        static const bool batched_t = !(getenv("MDH_MORPH_BT") && atoi(getenv("MDH_MORPH_BT")) == 0);
        if (batched_t)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_morph_close_t<false>), dim3((unsigned)std::min(K * wpf, 16384)), dim3(64), 0, post, sl.d_bbits, sl.d_worklist, sl.d_workcnt, c->d_act,
                               sl.d_nzlist, sl.d_nzcount, g, dense ? sl.d_dense : nullptr, wpf);
        else
        hipLaunchKernelGGL(k_morph_close, dim3(4096), dim3(64), 0, post, sl.d_bbits, sl.d_worklist, sl.d_workcnt, c->d_act,
                           sl.d_nzlist, sl.d_nzcount, g, dense ? sl.d_dense : nullptr);
        if (dense)
        {
            static const int dense_grid = getenv("MDH_DENSE_GRID") ? std::max(64, atoi(getenv("MDH_DENSE_GRID"))) : 8192;  // (2048 waves: 1.9 ms per 128 dense 8K frames, 8192: 1.5 -- a wave is a chain of 22 dependent row loads)
            hipLaunchKernelGGL(k_morph_close_dense, dim3(dense_grid), dim3(64), 0, post, sl.d_bbits, sl.d_dense, c->d_act, sl.d_nzlist, sl.d_nzcount, g);
        }
        }
        }
        span_end(c, post);
        EV_REC(c, sl.ev_close, post);
        // the dynamic mask looks n-1 frames back: the previous pass's binaries (written on another post stream when the
        // Hough sets alternate) must be in the ring; older passes are ordered by the streams themselves
        if (c->prev_slot >= 0 && c->prev_slot != (int)(&sl - c->slot))
            EV_WAIT(c, post, c->slot[c->prev_slot].ev_close);
        span_begin(c, KK_DY_APPLY, post);
        if (!(skip_stages() & 4))
        {
            // (k_dy_sparse<16>: sixteen history words per round trip for the long windows -- measured SLOWER, 0.29 -> 0.40 ms per
            // 4000-frame pass at n = 25, gpurun_out/r06_25: most listed words are done after the first four, and the sixteen loads
            // cost the kernel its place beside the window waves; MDH_DY_BATCH16=1 for A/B)
            static const bool batch16 = getenv("MDH_DY_BATCH16") && atoi(getenv("MDH_DY_BATCH16")) != 0;
            if (c->n > 16 && batch16)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dy_sparse<16>), dim3(c->lean && K == 1 ? 64 : 2048), dim3(256), 0, post, c->d_act, sl.d_dst, sl.d_nzlist,
                                   sl.d_nzcount, sl.d_rowcount, sl.d_litcount, sl.d_bbox, g, c->cfg.dy_mask);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_dy_sparse<4>), dim3(c->lean && K == 1 ? 64 : 2048), dim3(256), 0, post, c->d_act, sl.d_dst, sl.d_nzlist,
                                   sl.d_nzcount, sl.d_rowcount, sl.d_litcount, sl.d_bbox, g, c->cfg.dy_mask);
        }
        span_end(c, post);
    } else {
        static const bool classic_rb = !(getenv("MDH_CLASSIC_TB") && atoi(getenv("MDH_CLASSIC_TB")) == 0);
        span_begin(c, KK_MORPH, post);
        if (sl.classic_sparse) {
            const size_t bm_dwords = (nb * c->WW + 31) / 32;
            const int wpf = bm_dwords > 256 ? 4 : (bm_dwords > 64 ? 2 : 1);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(k_morph_select_touched<false>), dim3((unsigned)((K + 4 / wpf - 1) / (4 / wpf))), dim3(256), 0, post,
                               sl.d_bandflag, sl.d_worklist, sl.d_workcnt, g, wpf, 1);
            hipLaunchKernelGGL(k_classic_close_t, dim3((unsigned)std::min(K * wpf, 16384)), dim3(64), 0, post, (const ActWord*)sl.d_cplanes,
                               (const unsigned*)sl.d_worklist, (const unsigned*)sl.d_workcnt, sl.d_dst, sl.d_rowcount, sl.d_litcount, sl.d_bbox,
                               sl.d_nzlist, sl.d_nzcount, g, wpf, (unsigned long long*)(sl.d_nzcount + 2));
        } else
        if (classic_rb)
            hipLaunchKernelGGL(k_classic_stencil_rb, dim3((unsigned)((nb * c->WW + 255) / 256), K), dim3(256), 0, post, sl.d_planes, sl.d_dst,
                               sl.d_rowcount, sl.d_litcount, sl.d_bbox, g);
        else
            hipLaunchKernelGGL(k_classic_stencil, dim3((words + 255) / 256, K), dim3(256), 0, post, sl.d_planes, sl.d_dst,
                               sl.d_rowcount, sl.d_litcount, sl.d_bbox, g);
        span_end(c, post);
    }
    EV_REC(c, sl.ev_pix, post);
    return MDH_OK;
}

// detect() from the binary image onwards: adaptive gap, HoughLinesP, result packing.  Runs on the Hough streams of the
// set run_pixels (or mdh_hough_planes) queued the slot's planes and counters on; starts after sl.ev_pix.
static int run_hough(mdh_ctx* c, Slot& sl, int K) {
    HoughSet& hs = c->hset[c->pass_seq % c->nhset];
    // (one stream per pass: the pass's own stream; the set index is the chain stream's index -- both pass_seq mod NHSET)
    hipStream_t hstream[2] = {c->lean ? c->nstream : (c->one_stream ? c->stream : hs.stream[0]), c->lean ? c->nstream : (c->one_stream ? c->stream : hs.stream[1])};
    HoughParams P{};
    P.H = c->H; P.W = c->W; P.WW = c->WW; P.numrho = c->numrho; P.half = (c->numrho - 1) / 2;
    P.threshold = c->cfg.hough_threshold;
    P.line_length = c->cfg.hough_min_len;  // cvRound(int) == int
    P.max_gap = c->cfg.hough_max_gap;
    P.fixed_gap = c->cfg.detector == MDH_DET_CLASSIC;
    P.serial = c->hough_serial;
    P.no_lds_acc = c->hough_global_acc;
    // MDH_HOUGH_DEBUG_PASS=k: only the k-th pass records (to look at a pass in the middle of a pipelined run)
    static const long long dbg_pass = getenv("MDH_HOUGH_DEBUG_PASS") ? atoll(getenv("MDH_HOUGH_DEBUG_PASS")) : -1;
    const bool dbg_on = c->d_hdbg && (dbg_pass < 0 || c->pass_seq == dbg_pass);
    P.dbg = dbg_on ? c->d_hdbg : nullptr;
    P.dbg_phase = dbg_on ? c->d_hdbg + (size_t)c->Kmax * 8 : nullptr;
    P.mask_area = c->cfg.mask_area;
    P.dst_bits = sl.d_dst; P.work_bits = hs.d_work; P.rowcount = sl.d_rowcount; P.litcount = sl.d_litcount;
    P.accum = hs.d_accum; P.nz_orig = hs.d_nz_orig; P.nz_work = hs.d_nz_work; P.nz_cap = (long long)c->HW;
    P.out_lines = sl.d_lines; P.out_nraw = sl.d_nraw; P.out_gap = sl.d_gap; P.trig = c->d_trig;
    P.bbox = sl.d_bbox;
    P.tier_list = hs.d_tier_list; P.tier_count = hs.d_tier_count; P.K = K;
    // (lean mode: one three-wave launch with the large LDS takes every frame that fits; an empty launch less in the chain)
    P.lds_small = c->lean ? c->lds_large : c->lds_small; P.lds_large = c->lds_large;
    P.lds_tiny = c->lean ? 0 : c->lds_tiny;
    P.all_one_wave = c->hough_one_wave;
    P.perm = c->d_perm;
    P.rnd = c->d_rnd;
    P.perm_big = hs.d_perm_big;
    P.mute = (int)std::max<int64_t>(0, std::min<int64_t>(K, c->mute_until - sl.t_base));
    P.acc_mode = c->cfg.hough_mode == MDH_HOUGH_ACCUMULATOR;
    P.band_routing = c->hough_bands && !c->lean;  // (the per-frame protocol's one launch has the whole LDS anyway)
    P.list_cap = HOUGH_MAX_LIST; P.cell_cap = 61440;  // the one-wave kernel's LDS
    // Routing first (k_hough_classify: adaptive gap, and which launch replays each frame), then the launches side by side:
    // the small-LDS three-wave launch on stream [0]; the large-LDS launch, the hybrid launch (pooled global accumulator) and
    // the one-wave launch (global point lists) on stream [1].  Every launch walks a device-built frame list, so an empty tier
    // costs a launch of workgroups that leave at once.  The set's previous user queued its work on these same streams and
    // stream [0] waited for its stream [1] before packing, so the pooled buffers and the lists are free by stream order.
    EV_WAIT(c, hstream[0], sl.ev_pix);
    bool large_split = false;
    span_begin(c, KK_HOUGH, hstream[0]);
    static const bool skip_hough = getenv("MDH_SKIP_HOUGH") != nullptr || (skip_stages() & 8);  // timing experiments only: results are wrong
    if (c->lean && K == 1 && !skip_hough) {
        // the per-frame protocol: routing + replay + packing in one launch (k_hough_single), then the one result copy
        P.lds_small = P.lds_large;
        P.lds_bytes = P.lds_large;
        const size_t nb = (size_t)(c->H + MORPH_RB - 1) / MORPH_RB;
        sl.seq_no += 1;
        if (sl.seq_no == 0) sl.seq_no = 1;
        sl.seq_armed = c->lean_poll;
        SingleResult R{sl.h_off, sl.h_nraw, sl.h_litcount, sl.h_gap, sl.h_fscal, (int4*)sl.h_lines, sl.d_fscal,
                       (c->lean_direct && !c->lean_due) ? c->d_state : nullptr, sl.seq_armed ? sl.h_seq : nullptr, sl.seq_no,
                       c->mail_next && mail_clean_on() ? 0 : 1, sl.d_bandflag, (unsigned)(nb * c->WW), sl.d_sums};  // (mail_next: the waiting window kernel cleans up)
        sl.lean_clean = true;
        const size_t lds_single = std::max<size_t>(P.lds_bytes, (P.list_cap + P.cell_cap / 2 + 64) * sizeof(unsigned));
        if (P.dbg_phase) hipLaunchKernelGGL(k_hough_single<true>, dim3(1), dim3(HOUGH3_THREADS), lds_single, hstream[0], P, R);
        else hipLaunchKernelGGL(k_hough_single<false>, dim3(1), dim3(HOUGH3_THREADS), lds_single, hstream[0], P, R);
        span_end(c, hstream[0]);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(sl.ev_done, hstream[0]));
        sl.host_complete = true;
        hs.last_slot = (int)(&sl - c->slot);
        c->prev_slot = (int)(&sl - c->slot);
        c->pass_seq += 1;
        return MDH_OK;
    }
    if (skip_hough) {
        HIPCHK(hipMemsetAsync(sl.d_nraw, 0, sizeof(int) * K, hstream[0]));
        EV_REC(c, hs.ev_big, hstream[0]);
    } else {
        // (hs.d_tier_count is zero: allocated so, and k_pack_lines of the set's previous pass left it so)
        hipLaunchKernelGGL(k_hough_classify, dim3((K + 3) / 4), dim3(256), 0, hstream[0], P);
        EV_REC(c, hs.ev_cls, hstream[0]);
        EV_WAIT(c, hstream[1], hs.ev_cls);
        const int Kmax = c->Kmax;
        static const bool queue_on = getenv("MDH_HOUGH_QUEUE") && atoi(getenv("MDH_HOUGH_QUEUE")) != 0;  // (measured: no gain, gpurun_out/r06_34)
        auto tier = [&](int t) {
            P.flist = hs.d_tier_list + (size_t)t * K;
            P.fcount = hs.d_tier_count + t;
            // (queue heads behind the counts: [HOUGH_TIERS] k_hough3_big, + 1 small launch, + 2 tiny launch, + 3 large launch on its own)
            P.fqueue = (queue_on && (t == 0 || t == 4 || t == 1)) ? hs.d_tier_count + HOUGH_TIERS + (t == 0 ? 1 : (t == 4 ? 2 : 3)) : nullptr;
        };
        P.lds_bytes = c->lds_large;
        auto launch_h3 = [&](bool hybrid, int grid, hipStream_t st) {
            const bool tm = P.dbg_phase != nullptr;
            if (hybrid) {
                if (tm) hipLaunchKernelGGL(k_hough3_hybrid<true>, dim3(grid), dim3(HOUGH3_THREADS), P.lds_bytes, st, P);
                else hipLaunchKernelGGL(k_hough3_hybrid<false>, dim3(grid), dim3(HOUGH3_THREADS), P.lds_bytes, st, P);
            } else {
                if (tm) hipLaunchKernelGGL(k_hough3<true>, dim3(grid), dim3(HOUGH3_THREADS), P.lds_bytes, st, P);
                else hipLaunchKernelGGL(k_hough3<false>, dim3(grid), dim3(HOUGH3_THREADS), P.lds_bytes, st, P);
            }
        };
        static const int grid_l = getenv("MDH_HOUGH_GRID_LARGE") ? atoi(getenv("MDH_HOUGH_GRID_LARGE")) : 256;
        static const int grid_s = getenv("MDH_HOUGH_GRID_SMALL") ? atoi(getenv("MDH_HOUGH_GRID_SMALL")) : 1024;
        // The hybrid frames are the longest replays of a pass (a 2000-point 4K streak: 0.6-0.8 ms against 0.1-0.3 ms for a
        // frame that lives in LDS), so their launch goes first and the large-LDS launch runs BESIDE it on a stream of its own
        // instead of in front of it: the stage then lasts as long as its longest frame, not as the sum of the tiers' tails.
        // (MDH_HOUGH_SPLIT=0: round-2a order, large -> hybrid -> one-wave on one stream.)
        static const bool split_env = !(getenv("MDH_HOUGH_SPLIT") && atoi(getenv("MDH_HOUGH_SPLIT")) == 0);
        // MDH_HOUGH_BIG=0: hybrid and large-LDS frames as two launches side by side (round 2 / early round 3) instead of one launch
        // with a work queue (k_hough3_big)
        static const bool big = !(getenv("MDH_HOUGH_BIG") && atoi(getenv("MDH_HOUGH_BIG")) == 0);
        const bool split = split_env && !(big && !c->lean);
        hipStream_t lstream = (split && !c->one_stream) ? hs.stream[2] : hstream[1];
        if (!c->lean && !split && !big) {
            tier(1);
            launch_h3(false, std::min(K, grid_l), hstream[1]);
        }
        // Short streaks, four frames per CU (tier 4).  Up to round 5 this launch went IN FRONT of the whole-LDS launches of its stream
        // (k_hough3_big, k_hough_ppht: at such frame sizes usually empty): an empty launch of 256 workgroups that each want a CU's
        // whole LDS then waits for the tiny launch's last workgroups to leave -- 50-170 us at the end of a 960x540 pass's Hough stage,
        // which is the pipeline's critical chain there (gpurun_out/r06_36).  Round 6: behind them -- the empty launches run through
        // while the LDS is still free (MDH_HOUGH_TINY_FIRST=1: the old order).
        static const bool tiny_first = getenv("MDH_HOUGH_TINY_FIRST") && atoi(getenv("MDH_HOUGH_TINY_FIRST")) != 0;
        auto launch_tiny = [&]() {
            tier(4);
            P.lds_bytes = P.lds_tiny;
            launch_h3(false, std::min(K, grid_s), hstream[1]);
            P.lds_bytes = c->lds_large;
        };
        if (P.lds_tiny > 0 && tiny_first) launch_tiny();
        tier(2);
        if (big && !c->lean) {
            const int gb = std::min(K, hs.pool);  // (a work queue: any grid is correct; capped at 8 or 1 it changes nothing at 1080p / 960x540, where the launch is empty)
            if (P.dbg_phase) hipLaunchKernelGGL(k_hough3_big<true>, dim3(gb), dim3(HOUGH3_THREADS), P.lds_bytes, hstream[1], P);
            else hipLaunchKernelGGL(k_hough3_big<false>, dim3(gb), dim3(HOUGH3_THREADS), P.lds_bytes, hstream[1], P);
        } else
        launch_h3(true, std::min(K, hs.pool), hstream[1]);
        tier(3);
        hipLaunchKernelGGL(k_hough_ppht, dim3(std::min(K, hs.pool_list)), dim3(64), (P.list_cap + P.cell_cap / 2 + 64) * sizeof(unsigned),
                           hstream[1], P);
        if (P.lds_tiny > 0 && !tiny_first) launch_tiny();
        EV_REC(c, hs.ev_big, hstream[1]);
        if (!c->lean && split) {
            EV_WAIT(c, lstream, hs.ev_cls);
            tier(1);
            launch_h3(false, std::min(K, grid_l), lstream);
            EV_REC(c, hs.ev_big2, lstream);
            large_split = true;
        }
        tier(0);
        P.lds_bytes = P.lds_small;
        launch_h3(false, std::min(K, grid_s), hstream[0]);
        (void)Kmax;
    }
    EV_WAIT(c, hstream[0], hs.ev_big);
    if (large_split) EV_WAIT(c, hstream[0], hs.ev_big2);
    span_end(c, hstream[0]);
    HIPCHK(hipGetLastError());
    // pack the segments and send the head of the packed array along with the scalars; collect_pass fetches the rest only
    // when a pass holds more than PACK_HEAD segments per frame on average
    hipLaunchKernelGGL(k_pack_lines, dim3(1), dim3(1024), 0, hstream[0], (const int4*)sl.d_lines, sl.d_nraw, K, MAX_LINES,
                       (int4*)sl.d_packed, sl.d_off, sl.d_litcount, sl.d_litcopy,
                       // the format cue: dense pass -> words of b with set bits (u64); sparse pass -> its number of dense frames
                       (sl.dense || sl.classic_sparse) ? (const unsigned*)(sl.d_nzcount + 2) : (const unsigned*)sl.d_dense,
                       skip_hough ? (int*)nullptr : hs.d_tier_count);
    // ONE copy: offsets, counts, gaps, frame scalars and the head of the packed segments
    HIPCHK(hipMemcpyAsync(sl.h_res, sl.d_res, sl.res_lines_off + sizeof(int) * 4 * (size_t)std::min(PACK_HEAD, MAX_LINES) * K,
                          hipMemcpyDeviceToHost, hstream[0]));
    HIPCHK(hipEventRecord(sl.ev_done, hstream[0]));
    hs.last_slot = (int)(&sl - c->slot);
    c->prev_slot = (int)(&sl - c->slot);
    c->pass_seq += 1;
    return MDH_OK;
}

static int run_detect(mdh_ctx* c, Slot& sl, int K) {
    sl.host_complete = false;
    const int rc = run_pixels(c, sl, K);
    return rc != MDH_OK ? rc : run_hough(c, sl, K);
}

// After a pass is queued: remember where the last n-1 frames live.  Frames that came from the host already sit in the ring
// (slot = frame number mod R).  Frames the caller owns in device memory are copied -- only the ones the next passes' windows
// reach back to -- into the history buffer: slots are handed out by a running counter, and (nslot+1)(n-1) of them are enough,
// because a frame leaves the history after n-1 newer copies and the nslot(n-1) copies that follow before its slot comes up
// again take at least nslot passes, by which time every pass that could still read it has been collected.
static int retire_frames(mdh_ctx* c, Slot& sl, int K, int on_device) {
    const int64_t t_base = sl.t_base;
    const int keep = c->n - 1;
    std::vector<const uint8_t*> nh(keep);
    for (int e = 0; e < keep; ++e) {
        // entry e of the new history = table entry K + e of this pass
        const int src = K + e;
        if (src < keep) {
            nh[e] = c->hist[src];
        } else if (!on_device) {
            nh[e] = c->d_ring + (size_t)((t_base + (src - keep)) % c->R) * c->HW;
        } else {
            nh[e] = nullptr;  // filled below
        }
    }
    if (on_device && keep > 0) {
        const int first = std::max(0, K - keep);  // first pass frame that stays in the history
        int i = first;
        while (i < K) {  // contiguous in the caller's buffer; split where the history buffer wraps
            const int64_t hs = c->hist_pos % c->R_hist;
            const int run = (int)std::min<int64_t>(K - i, c->R_hist - hs);
            HIPCHK(hipMemcpyAsync(c->d_hist + (size_t)hs * c->HW, sl.h_ptrtab[keep + i], (size_t)run * c->HW, hipMemcpyDeviceToDevice,
                                  c->nstream));
            for (int k = 0; k < run; ++k) nh[keep - (K - (i + k))] = c->d_hist + (size_t)(hs + k) * c->HW;
            c->hist_pos += run;
            i += run;
        }
    }
    c->hist.swap(nh);
    return MDH_OK;
}

// Take the next slot for a pass of K frames.  The slot's previous pass must have been collected.
static int slot_begin(mdh_ctx* c, int K, Slot*& out) {
    if (K < 1 || K > c->Kmax) { set_err("pass size must be 1..max_batch"); return MDH_E_ARG; }
    if (!c->lean_entry) mail_cancel(c);  // (mdh_update decides for itself whether the waiting window kernel gets its frame)
    if (c->in_flight >= c->nslot) { set_err("the maximum number of passes is already in flight: collect one first"); return MDH_E_STATE; }
    Slot& sl = c->slot[c->next_slot];
    if (sl.busy) { set_err("internal: slot still busy"); return MDH_E_STATE; }
    // the main stream may not overwrite this slot's buffers before the Hough stage that last read them is done
    // Consecutive passes alternate between two main streams, so the next window kernel fills the machine while the last
    // workgroups of the previous one drain (+4 % once the post stages had streams of their own; MDH_MAIN_STREAMS=1 for A/B).
    c->stream = (c->one_stream && !c->lean_entry) ? c->pass_stream[c->pass_seq % NHSET] : c->mstream[c->pass_seq % c->nmain];
    // (a per-frame pass runs on the staging stream alone, behind the slot's previous pass if that was a per-frame one too; the
    // event of a previous BATCHED pass -- recorded on a Hough stream -- is waited for by the staging stream instead: mdh_update)
    if (sl.used && !c->lean_entry) HIPCHK(hipStreamWaitEvent(c->stream, sl.ev_done, 0));
    if (sl.used && c->lean_entry && !sl.was_lean) HIPCHK(hipStreamWaitEvent(c->nstream, sl.ev_done, 0));
    c->prof = c->prof_every > 0 && c->pass_seq % c->prof_every == 0;
    sl.K = K;
    sl.t_base = c->t;
    sl.dense = false;
    if (c->dense_now && c->cfg.detector == MDH_DET_M3 && c->role != MDH_ROLE_HOUGH && K > 1) {
        if (!sl.d_bplain) {
            // First dense pass of the context: the plain buffers of ALL its slots at once.  Round 4 allocated one per slot as the
            // passes came by -- six allocations, each synchronising the device, spread over the first six dense passes (a timed
            // region behind three warm-up passes still paid three of them, times eight contexts in the row-band bench); and on a
            // box whose VRAM has not been handed out before, an allocation of gigabytes is slow (the driver wipes the pages it
            // hands out; a later process gets pages the driver wiped in the background when the earlier one exited) -- the
            // "first process on a fresh box runs the band bench at a quarter of the rate" of round 4.
            for (int k = 0; k < c->nslot; ++k) {
                Slot& o = c->slot[k];
                if (o.d_bplain) continue;
                if (hipMalloc((void**)&o.d_bplain, sizeof(u64) * (size_t)c->Kmax * c->words) != hipSuccess) {
                    (void)hipGetLastError();
                    o.d_bplain = nullptr;
                }
            }
            if (!sl.d_bplain) {  // no memory for the second format: stay with the sparse one
                c->dense_now = false;
                c->dense_cfg = 0;
            }
        }
        sl.dense = sl.d_bplain != nullptr;
    }
    sl.was_clean = sl.lean_clean;
    sl.lean_clean = false;
    sl.seq_armed = false;
    static const bool tag_pass = getenv("MDH_TAG_PASS") != nullptr;
    if (tag_pass) c->tag_lds = 1024u * (1 + (unsigned)((c->pass_seq + c->in_flight) % 3));
    c->cur_spans = &sl.spans;
    c->d_ptrtab = sl.d_ptrtab; c->d_jobs = sl.d_jobs; c->d_sums = sl.d_sums; c->d_jobres = sl.d_jobres; c->d_thr = sl.d_thr; c->d_wrec = sl.d_wrec;
    c->d_bandflag = sl.d_bandflag; c->d_bbits = sl.d_bbits;
    out = &sl;
    return MDH_OK;
}
static void slot_commit(mdh_ctx* c, Slot& sl) {
    sl.was_lean = c->lean_commit;
    sl.busy = true;
    sl.used = true;
    c->t += sl.K;
    c->next_slot = (c->next_slot + 1) % c->nslot;
    c->in_flight += 1;
}

static int poison(mdh_ctx* c, int rc) {
    c->poisoned = true;
    set_err(g_err + " -- the context's stream state is no longer consistent; destroy it");
    return rc;
}
#define CHECK_POISON(c, who)                                                                             \
    if ((c)->poisoned) {                                                                                 \
        set_err(who ": an earlier call failed half-way through a pass; the context must be destroyed");   \
        return MDH_E_STATE;                                                                              \
    }

// Wait for a pass's completion event.  A pass that contains a collective (row bands over RCCL) never completes when a peer has
// died; with a timeout set the wait gives up, poisons the context and says so -- the host then aborts its communicator, which
// releases the stuck kernels, and destroys the context.
static int wait_event(mdh_ctx* c, hipEvent_t ev, const char* who) {
    if (!(c->wait_timeout_ms > 0)) {
        HIPCHK(hipEventSynchronize(ev));
        return MDH_OK;
    }
    const double t0 = now_us();
    for (int spin = 0;; ++spin) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return MDH_OK;
        if (e != hipErrorNotReady) {
            set_err(std::string(who) + ": hipEventQuery: " + hipGetErrorString(e));
            return MDH_E_HIP;
        }
        (void)hipGetLastError();
        if (now_us() - t0 > c->wait_timeout_ms * 1e3) {
            c->poisoned = true;
            set_err(std::string(who) + ": the pass did not complete within " + std::to_string((long long)c->wait_timeout_ms) +
                    " ms (a collective whose peer is gone?) -- abort the communicator, then destroy the context");
            return MDH_E_TIMEOUT;
        }
        if (spin > 64) {
            timespec ts{0, 20000};
            nanosleep(&ts, nullptr);
        }
    }
}
extern "C" int mdh_set_wait_timeout_ms(mdh_ctx* c, double ms) {
    if (!c || ms < 0) { set_err("mdh_set_wait_timeout_ms: bad argument"); return MDH_E_ARG; }
    c->wait_timeout_ms = ms;
    return MDH_OK;
}

static int submit_pass(mdh_ctx* c, const uint8_t* frames, int K, int on_device) {
    Slot* sl = nullptr;
    int rc;
    const double t0 = now_us();
    host_log("submit begin", c->pass_seq);
    if ((rc = slot_begin(c, K, sl)) != MDH_OK) return rc;
    // staging stream: pointer table (built from the history BEFORE this pass), history copies, noise / thresholds;
    // main stream: pixel stages once the thresholds are there.  stage_frames advances the EMA schedule and retire_frames
    // the history: a failure from here on leaves the context inconsistent (poisoned).
    if ((rc = stage_frames(c, *sl, frames, K, on_device)) != MDH_OK) return poison(c, rc);
    if ((rc = retire_frames(c, *sl, K, on_device)) != MDH_OK) return poison(c, rc);
    if ((rc = run_noise(c, *sl, K)) != MDH_OK) return poison(c, rc);
    if ((rc = run_detect(c, *sl, K)) != MDH_OK) return poison(c, rc);
    slot_commit(c, *sl);
    c->host_us[0] += now_us() - t0;
    host_log("submit end", c->pass_seq - 1);
    return MDH_OK;
}

// The format of the passes queued from now on, from what the pass just collected reported (hysteresis; both formats give the
// same results): a sparse pass reports its number of DENSE FRAMES (at least half of the morphology blocks listed: k_morph_close),
// a dense pass the number of words of b with set bits.
static void format_cue(mdh_ctx* c, const Slot& sl) {
    const int K = sl.K;
    if (c->cfg.detector != MDH_DET_M3) {
        // ClassicDetector, sparse pass: the number of blocks its stencil was handed; more than a quarter of all of them -- the
        // threshold lets the noise through -- and the next 32 passes run the dense kernels (then the sparse format is tried again)
        if (sl.classic_sparse) {
            const unsigned long long listed = (unsigned long long)(unsigned)sl.h_off[K + 1] | ((unsigned long long)(unsigned)sl.h_off[K + 2] << 32);
            const double per_frame = (double)((c->H + MORPH_RB - 1) / MORPH_RB) * c->WW;
            if ((double)listed > 0.25 * per_frame * K) c->classic_dense_passes = 32;
        }
        return;
    }
    if (!sl.dense) {
        const double share = (double)(unsigned)sl.h_off[K + 1] / (double)K;  // frames whose binary was dense
        c->last_density = share;
        if (c->dense_cfg < 0 && !c->dense_now && share >= 0.5) c->dense_now = true;
    } else {
        const unsigned long long nzw = (unsigned long long)(unsigned)sl.h_off[K + 1] | ((unsigned long long)(unsigned)sl.h_off[K + 2] << 32);
        c->last_density = (double)nzw / ((double)K * (double)c->words);
        if (c->dense_cfg < 0 && c->dense_now && c->last_density < c->dense_leave) c->dense_now = false;
    }
}

// Wait for the oldest pass in flight and hand its results to the caller.
static int collect_pass(mdh_ctx* c, mdh_frame_info* info, int32_t* raw_lines, int32_t* nms_lines, double* nonline, int* K_out) {
    if (c->in_flight <= 0) { set_err("no pass in flight"); return MDH_E_STATE; }
    Slot& sl = c->slot[c->oldest];
    const int K = sl.K;
    const double t0 = now_us();
    host_log("collect begin", c->pass_seq - c->in_flight);
    bool seen = false;
    if (sl.seq_armed && sl.host_complete && sl.spans.empty()) {
        // the kernel stores this word last (system-scope release) into the pinned result block: seeing it is seeing the results,
        // a few microseconds before the completion signal of the kernel would wake the host.  Bounded: after ~2 ms the event decides.
        volatile unsigned* flag = sl.h_seq;
        const double t_spin = now_us();
        for (int spin = 0; !seen; ++spin) {
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == sl.seq_no) seen = true;
            else if ((spin & 1023) == 1023 && now_us() - t_spin > 2000) break;
        }
    }
    if (!seen) {
        const int wrc = wait_event(c, sl.ev_done, "collect");
        if (wrc != MDH_OK) return wrc;
    }
    host_log("collect woke", c->pass_seq - c->in_flight);
    const double t1 = now_us();
    spans_collect(c, sl);
    // the head of the packed segment array came with the scalars; fetch the tail if there is one
    const size_t head = (size_t)std::min(PACK_HEAD, MAX_LINES) * K, total = (size_t)sl.h_off[K];
    if (total > head && !sl.host_complete) {
        HIPCHK(hipMemcpyAsync(sl.h_lines + 4 * head, sl.d_packed + 4 * head, sizeof(int) * 4 * (total - head), hipMemcpyDeviceToHost,
                              c->cstream));
        HIPCHK(hipStreamSynchronize(c->cstream));
    }
    const double t2 = now_us();
    std::vector<int32_t> tmp_l(nms_lines ? 0 : 4 * MAX_LINES);  // (only a caller that does not want the NMS output needs somewhere to put it)
    std::vector<double> tmp_p(nonline ? 0 : MAX_LINES);
    for (int i = 0; i < K; ++i) {
        // (the caller's arrays are [K][MAX_LINES] strided: a frame with segments touches three cold lines 8-16 KB apart; asked for a
        // few frames ahead they arrive side by side -- the unpacking of a 4000-frame pass is otherwise a chain of 4000 cache misses)
        if (i + 6 < K && sl.h_nraw[i + 6] > 0) {
            if (raw_lines) __builtin_prefetch(raw_lines + (size_t)(i + 6) * MAX_LINES * 4, 1);
            if (nms_lines) __builtin_prefetch(nms_lines + (size_t)(i + 6) * MAX_LINES * 4, 1);
            if (nonline) __builtin_prefetch(nonline + (size_t)(i + 6) * MAX_LINES, 1);
        }
        const int n_raw = sl.h_nraw[i];
        const int stored = std::min(n_raw, MAX_LINES);
        const int32_t* L = sl.h_lines + (size_t)sl.h_off[i] * 4;
        if (raw_lines) memcpy(raw_lines + (size_t)i * MAX_LINES * 4, L, sizeof(int32_t) * 4 * stored);
        int n_nms = 0;
        if (n_raw > 0 && n_raw <= 500) {  // NUM_LINES_TOOMUCH (Detector.py:30, 358-360)
            n_nms = mdh_lineset_nms(L, stored, nms_lines ? nms_lines + (size_t)i * MAX_LINES * 4 : tmp_l.data(),
                                    nonline ? nonline + (size_t)i * MAX_LINES : tmp_p.data());
        }
        if (info) {
            mdh_frame_info& o = info[i];
            o.n_raw = n_raw;
            o.n_nms = n_nms;
            o.lit_count = sl.h_litcount[i];
            o.bi_threshold = sl.h_fscal[i].thr;
            o.bi_threshold_float = sl.h_fscal[i].thr_f;
            o.snr = sl.h_fscal[i].snr;
            o.sigma = sl.h_fscal[i].sigma;
            o.dst_sum = (double)sl.h_litcount[i] / c->cfg.mask_area * 100;  // Detector.py:342
            o.hough_gap = sl.h_gap[i];
        }
    }
    if (!sl.host_complete && c->role != MDH_ROLE_HOUGH) format_cue(c, sl);
    c->host_us[1] += t1 - t0;
    c->host_us[2] += t2 - t1;
    c->host_us[3] += now_us() - t2;
    c->host_n += 1;
    sl.busy = false;
    c->last_slot = c->oldest;
    c->oldest = (c->oldest + 1) % c->nslot;
    c->in_flight -= 1;
    if (K_out) *K_out = K;
    return MDH_OK;
}

extern "C" int mdh_submit_batch(mdh_ctx* c, const uint8_t* frames, int K, int on_device) {
    if (!c || !frames) { set_err("mdh_submit_batch: null argument"); return MDH_E_ARG; }
    if (c->role != MDH_ROLE_FULL) { set_err("mdh_submit_batch: not a MDH_ROLE_FULL context"); return MDH_E_STATE; }
    if (c->pending) { set_err("mdh_submit_batch: a frame from mdh_update is still pending"); return MDH_E_STATE; }
    CHECK_POISON(c, "mdh_submit_batch");
    HIPCHK(hipSetDevice(c->device));
    return submit_pass(c, frames, K, on_device);
}

extern "C" int mdh_collect_batch(mdh_ctx* c, mdh_frame_info* info, int32_t* raw_lines, int32_t* nms_lines, double* nonline) {
    if (!c) { set_err("mdh_collect_batch: null context"); return MDH_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    int K = 0;
    const int rc = collect_pass(c, info, raw_lines, nms_lines, nonline, &K);
    return rc == MDH_OK ? K : rc;
}

// Per-frame protocol, fast path: may the window kernel read this frame from the caller's buffer itself?  Yes when the buffer
// is host memory a kernel can reach (hipHostMalloc / hipHostRegister: mdh_host_alloc), the slot is clean (its last pass was a
// per-frame one), a wide-load window kernel applies (16-byte aligned rows of 64-pixel words) and no sigma is due for the frame
// (Detector.py:82,87: one frame in interval * n) -- then the threshold is the running one and the noise chain has nothing to do.
// Returns the device-visible address of the frame, or null (the general path: DMA copy, header copy, prepare, noise chain).
static const uint8_t* lean_direct_source(mdh_ctx* c, const Slot& sl, const uint8_t* frame) {
    if (!c->lean_direct_on || !sl.was_clean || !sl.h_hdr_dev) return nullptr;
    if (c->cfg.detector != MDH_DET_M3 || c->W % 64 != 0 || c->force_generic || c->force_bytes || c->n < 2 || (c->n > 16 && !long_window_wide(c->n))) return nullptr;
    const int64_t t = c->t + 1;
    const int64_t std_interval = (int64_t)c->cfg.interval * c->n;
    c->lean_due = t > c->n && std_interval > 0 && t % std_interval == 0;
    if (t <= c->n || (c->lean_due && !c->lean_due_on)) { c->lean_due = false; return nullptr; }
    // sums the caller supplied for this frame (mdh_inject_sums after mdh_seek) are applied by the general path's noise chain only
    if (c->lean_due && !c->injected.empty()) { c->lean_due = false; return nullptr; }
    auto usable = [&](uintptr_t dp) -> const uint8_t* {
        if (dp & 15) return nullptr;
        for (int e = 0; e < c->n - 1; ++e)
            if ((uintptr_t)c->hist[e] & 15) return nullptr;
        return (const uint8_t*)dp;
    };
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (const HostPool& hp : g_pools)
            if (hp.device == c->device && (uintptr_t)frame >= hp.base && (uintptr_t)frame + c->HW <= hp.base + hp.bytes)
                return usable(hp.dev + ((uintptr_t)frame - hp.base));
    }
    // (Up to round 5 the runtime was asked as well -- hipPointerGetAttributes: memory the caller pinned with hipHostMalloc /
    // hipHostRegister of its own was read in place.  The runtime gives the same answer for a range that was registered once and
    // freed without being unregistered, and a frame read through such a stale mapping is a GPU memory fault that aborts the process
    // (tools/stress_pre_run.py --mode stale): only ranges this library knows -- mdh_host_alloc, mdh_host_register,
    // its own auto-registrations -- are read in place now; a loader with pinned memory of its own registers it with mdh_host_register.)
    // A caller that hands over the same address again and again recycles a fixed set of buffers: the third time round the
    // buffer is registered (pinned and mapped, ~0.2 ms once) and from then on read in place like a pool of mdh_host_alloc.
    if (!c->auto_register_on || c->auto_registered >= AUTO_REG_MAX_BUFFERS) return nullptr;
    int hits = 0;
    for (auto& e : c->reg_seen)
        if (e.addr == (uintptr_t)frame) { hits = ++e.hits; break; }
    if (!hits) {
        c->reg_seen[c->reg_seen_next] = {(uintptr_t)frame, 1};
        c->reg_seen_next = (c->reg_seen_next + 1) % AUTO_REG_SEEN;
        return nullptr;
    }
    if (hits < AUTO_REG_HITS) return nullptr;
    void* dp = nullptr;
    if (hipHostRegister((void*)frame, c->HW, hipHostRegisterMapped) != hipSuccess ||
        hipHostGetDevicePointer(&dp, (void*)frame, 0) != hipSuccess || !dp) {
        (void)hipGetLastError();
        for (auto& e : c->reg_seen)
            if (e.addr == (uintptr_t)frame) e.hits = -(1 << 30);  // (cannot be registered: never asked again)
        return nullptr;
    }
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        g_pools.push_back({(uintptr_t)frame, (uintptr_t)c->HW, (uintptr_t)dp, c->device, 2, c});
    }
    c->auto_registered += 1;
    return usable((uintptr_t)dp);
}

// The reference's per-frame protocol (MetDetPy.py:197-198).  mdh_update queues the WHOLE pass of the frame -- copy, noise /
// threshold chain, pixel stages, Hough, result copy -- on one stream without a single cross-stream dependency, and returns
// as soon as the caller's buffer has been read (the reference copies it, utils.py:280; while the 2 MB travel over PCIe the
// host has queued everything else); mdh_detect waits for the pass and hands out the results.
extern "C" int mdh_update(mdh_ctx* c, const uint8_t* frame) {
    if (!c || !frame) { set_err("mdh_update: null argument"); return MDH_E_ARG; }
    if (c->role != MDH_ROLE_FULL) { set_err("mdh_update: not a MDH_ROLE_FULL context"); return MDH_E_STATE; }
    if (c->pending) { set_err("mdh_update: previous frame not yet consumed by mdh_detect"); return MDH_E_STATE; }
    if (c->in_flight) { set_err("mdh_update: collect the submitted passes first"); return MDH_E_STATE; }
    CHECK_POISON(c, "mdh_update");
    // a caller that comes back within a few dozen microseconds of the last results is in a tight loop (the reference's, fed from
    // memory); one that decodes video between frames is not, and gets no kernel spinning on its behalf
    c->tight_loop = c->t_detect_ret > 0 && now_us() - c->t_detect_ret < c->mail_gap_us;
    HIPCHK(hipSetDevice(c->device));
    Slot* sl = nullptr;
    int rc;
    c->lean_entry = true;
    rc = slot_begin(c, 1, sl);
    c->lean_entry = false;
    if (rc != MDH_OK) return rc;
    c->lean = true;
    c->stream = c->nstream;
    c->lean_due = false;
    c->mail_used = false;
    c->lean_direct = lean_direct_source(c, *sl, frame);
    if (!c->lean_direct) c->lean_due = false;
    if (!c->lean_direct || c->lean_due || c->prof) mail_cancel(c);  // copies / noise launches / timing events come first on the staging stream
    // Will the next frame's window kernel be queued behind this frame (decided here: k_hough_single then leaves the slot's
    // clean-up to it)?  This frame takes the direct path, the caller is in a tight loop, and no sigma is due for the next frame
    // (a due frame's noise chain runs in front of its window kernel).
    c->mail_next = false;
    if (c->lean_direct && c->mail_on && c->tight_loop && c->lean_poll && c->prof_every == 0 && !c->prof && c->cfg.detector == MDH_DET_M3) {
        const int64_t t_next = c->t + 2, std_interval = (int64_t)c->cfg.interval * c->n;  // (c->t counts the frames before this one)
        c->mail_next = !(std_interval > 0 && t_next % std_interval == 0);
    }
    rc = stage_frames(c, *sl, frame, 1, 0);
    if (rc == MDH_OK) rc = retire_frames(c, *sl, 1, 0);
    if (rc == MDH_OK && !c->lean_direct) rc = run_noise(c, *sl, 1);
    if (rc == MDH_OK && c->lean_direct && c->lean_due) rc = run_noise_direct(c, *sl);
    if (rc == MDH_OK) rc = run_detect(c, *sl, 1);
    c->lean = false;
    c->lean_due = false;
    const bool direct = c->lean_direct != nullptr;
    c->lean_direct = nullptr;
    if (rc != MDH_OK) return poison(c, rc);
    (direct ? c->direct_frames : c->copied_frames) += 1;
    c->lean_commit = true;
    slot_commit(c, *sl);
    c->lean_commit = false;
    // The next frame's window kernel goes behind this frame's last launch (k_stack_threshold_mail) when the next frame can be
    // expected to take the same path: this one did, and no sigma is due for the next (a due frame's noise chain runs in front of
    // its window kernel).  Queued here, its launch costs the host 3 us of the 45 the frame needs to cross the bus.
    if (c->mail_next) {
        const size_t nbk = (size_t)(c->H + MORPH_RB - 1) / MORPH_RB;
        MailClean cl{sl->d_rowcount, sl->d_litcount, sl->d_bbox, sl->d_sums, sl->d_bandflag, (unsigned)(nbk * c->WW), c->H};
        if (!mail_clean_on()) cl.blockcnt = nullptr;
        mail_prelaunch(c, cl);
        if (!c->mail_armed && mail_clean_on()) {  // (could not be queued: the clean-up k_hough_single skipped, as plain fills)
            HIPCHK(hipMemsetAsync(sl->d_bandflag, 0, sizeof(unsigned) * nbk * c->WW, c->nstream));
            HIPCHK(hipMemsetAsync(sl->d_rowcount, 0, sizeof(int) * c->H, c->nstream));
            HIPCHK(hipMemsetAsync(sl->d_litcount, 0, sizeof(int), c->nstream));
            HIPCHK(hipMemsetAsync(sl->d_bbox, 0, sizeof(int) * 4, c->nstream));
            if (sl->d_sums) HIPCHK(hipMemsetAsync(sl->d_sums, 0, sizeof(long long) * 2, c->nstream));
        }
        c->mail_next = false;
    }
    // the caller may reuse `frame` after return
    bool read_seen = false;
    if (direct && c->lean_poll) {
        const double t_spin = now_us();
        for (int spin = 0; !read_seen; ++spin) {
            if (__atomic_load_n(c->h_read_flag, __ATOMIC_ACQUIRE) == c->read_seq) read_seen = true;
            else if ((spin & 1023) == 1023 && now_us() - t_spin > 2000) break;
        }
        if (!read_seen) HIPCHK(hipStreamSynchronize(c->nstream));  // (never expected: the whole chain instead of the word)
    } else {
        HIPCHK(hipEventSynchronize(c->ev_copy));
    }
    c->pending = true;
    return MDH_OK;
}

extern "C" int mdh_detect(mdh_ctx* c, mdh_frame_info* info, int32_t* raw_lines, int32_t* nms_lines, double* nonline) {
    if (!c) { set_err("mdh_detect: null context"); return MDH_E_ARG; }
    if (!c->pending) { set_err("mdh_detect: call mdh_update first"); return MDH_E_STATE; }
    CHECK_POISON(c, "mdh_detect");
    HIPCHK(hipSetDevice(c->device));
    c->pending = false;
    const int rc = collect_pass(c, info, raw_lines, nms_lines, nonline, nullptr);
    c->t_detect_ret = now_us();
    return rc;
}

extern "C" int mdh_process_batch(mdh_ctx* c, const uint8_t* frames, int K, int on_device, mdh_frame_info* info,
                                 int32_t* raw_lines, int32_t* nms_lines, double* nonline) {
    if (!c || !frames || K < 0) { set_err("mdh_process_batch: bad argument"); return MDH_E_ARG; }
    if (c->role != MDH_ROLE_FULL) { set_err("mdh_process_batch: not a MDH_ROLE_FULL context"); return MDH_E_STATE; }
    if (c->pending) { set_err("mdh_process_batch: a frame from mdh_update is still pending"); return MDH_E_STATE; }
    if (c->in_flight) { set_err("mdh_process_batch: collect the submitted passes first"); return MDH_E_STATE; }
    CHECK_POISON(c, "mdh_process_batch");
    HIPCHK(hipSetDevice(c->device));
    // chunks of at most max_batch frames, up to nslot in flight: later chunks are queued before chunk j is collected
    int submitted = 0, collected = 0;
    while (collected < K) {
        int rc;
        while (submitted < K && c->in_flight < c->nslot) {
            const int k = std::min(K - submitted, c->Kmax);
            if ((rc = submit_pass(c, frames + (size_t)submitted * c->HW, k, on_device)) != MDH_OK) return rc;
            submitted += k;
        }
        int k = 0;
        if ((rc = collect_pass(c, info ? info + collected : nullptr,
                               raw_lines ? raw_lines + (size_t)collected * MAX_LINES * 4 : nullptr,
                               nms_lines ? nms_lines + (size_t)collected * MAX_LINES * 4 : nullptr,
                               nonline ? nonline + (size_t)collected * MAX_LINES : nullptr, &k)) != MDH_OK)
            return rc;
        collected += k;
    }
    return MDH_OK;
}

// ------------------------------------------------------------------------------------------------ row-band tiling
static int xchg_reserve(mdh_ctx* c, size_t words) {
    if (c->xchg_words >= words) return MDH_OK;
    if (c->d_xchg) HIPCHK(hipFree(c->d_xchg));
    c->d_xchg = nullptr;
    c->xchg_words = 0;
    HIPCHK(hipMalloc((void**)&c->d_xchg, words * sizeof(u64)));
    c->xchg_words = words;
    return MDH_OK;
}

extern "C" int mdh_band_noise(mdh_ctx* c, const uint8_t* frames, int K, int on_device, int64_t* sums, int* n_jobs) {
    if (!c || !frames || !sums || !n_jobs) { set_err("mdh_band_noise: null argument"); return MDH_E_ARG; }
    if (c->role != MDH_ROLE_BAND) { set_err("mdh_band_noise: not a MDH_ROLE_BAND context"); return MDH_E_STATE; }
    if (c->band_slot) { set_err("mdh_band_noise: the previous pass still waits for mdh_band_pixels"); return MDH_E_STATE; }
    CHECK_POISON(c, "mdh_band_noise");
    HIPCHK(hipSetDevice(c->device));
    Slot* sl = nullptr;
    int rc, nj = 0;
    if ((rc = slot_begin(c, K, sl)) != MDH_OK) return rc;
    if ((rc = stage_frames(c, *sl, frames, K, on_device)) != MDH_OK) return poison(c, rc);
    if ((rc = retire_frames(c, *sl, K, on_device)) != MDH_OK) return poison(c, rc);
    if ((rc = noise_prepare(c, *sl, K, &nj)) != MDH_OK) return poison(c, rc);
    if (nj) HIPCHK(hipMemcpyAsync(sums, c->d_sums, sizeof(long long) * 2 * nj, hipMemcpyDeviceToHost, c->nstream));
    HIPCHK(hipStreamSynchronize(c->nstream));
    c->band_slot = sl;
    c->band_thr_done = false;
    c->band_nj = nj;
    *n_jobs = nj;
    return MDH_OK;
}

extern "C" int mdh_band_pixels(mdh_ctx* c, const int64_t* sums_total, double roi_px_total, int own_row0, int own_rows,
                               int rows_stride, int nsplit, uint64_t* dst_rows, int out_on_device, mdh_frame_info* info) {
    if (!c || !sums_total || !dst_rows) { set_err("mdh_band_pixels: null argument"); return MDH_E_ARG; }
    if (c->role != MDH_ROLE_BAND) { set_err("mdh_band_pixels: not a MDH_ROLE_BAND context"); return MDH_E_STATE; }
    if (!c->band_slot) { set_err("mdh_band_pixels: call mdh_band_noise first"); return MDH_E_STATE; }
    CHECK_POISON(c, "mdh_band_pixels");
    if (own_row0 < 0 || own_rows < 1 || own_row0 + own_rows > c->H || rows_stride < own_rows || nsplit < 1 || !(roi_px_total > 0)) {
        set_err("mdh_band_pixels: bad owned rows / stride / split / roi_px_total");
        return MDH_E_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    Slot& sl = *c->band_slot;
    const int K = sl.K, nj = c->band_nj;
    int rc;
    if (nj) HIPCHK(hipMemcpyAsync(c->d_sums, sums_total, sizeof(long long) * 2 * nj, hipMemcpyHostToDevice, c->nstream));
    if ((rc = noise_finish(c, sl, K, nj, roi_px_total)) != MDH_OK) return rc;
    HoughSet& hs = c->hset[c->pass_seq % c->nhset];
    hipStream_t post = post_stream(c, c->pass_seq);
    if ((rc = run_pixels(c, sl, K)) != MDH_OK) return rc;
    // owned rows -> exchange buffer in all-to-all send order
    const int Kp = (K + nsplit - 1) / nsplit;
    const size_t out_words = (size_t)nsplit * Kp * rows_stride * c->WW;
    u64* dev_out = (u64*)dst_rows;
    if (!out_on_device) {
        if ((rc = xchg_reserve(c, out_words)) != MDH_OK) return rc;
        dev_out = c->d_xchg;
    }
    HIPCHK(hipMemsetAsync(dev_out, 0, out_words * sizeof(u64), post));
    hipLaunchKernelGGL(k_export_rows, dim3((own_rows * c->WW + 255) / 256, K), dim3(256), 0, post, sl.d_dst, dev_out, K, c->H, c->WW,
                       own_row0, own_rows, rows_stride, nsplit, Kp);
    HIPCHK(hipGetLastError());
    if (!out_on_device) HIPCHK(hipMemcpyAsync(dst_rows, dev_out, out_words * sizeof(u64), hipMemcpyDeviceToHost, post));
    HIPCHK(hipMemcpyAsync(sl.h_fscal, sl.d_fscal, sizeof(FrameScalars) * K, hipMemcpyDeviceToHost, post));
    HIPCHK(hipMemcpyAsync(sl.h_off + K + 1, sl.dense ? (const void*)(sl.d_nzcount + 2) : (const void*)sl.d_dense, 2 * sizeof(unsigned),
                          hipMemcpyDeviceToHost, post));  // the format cue (see k_pack_lines)
    HIPCHK(hipEventRecord(sl.ev_done, post));
    const int si = (int)(&sl - c->slot);
    hs.last_slot = si;
    c->prev_slot = si;
    c->pass_seq += 1;
    slot_commit(c, sl);
    c->band_slot = nullptr;
    c->band_thr_done = false;
    HIPCHK(hipEventSynchronize(sl.ev_done));
    spans_collect(c, sl);
    format_cue(c, sl);
    if (info) {
        for (int i = 0; i < K; ++i) {
            mdh_frame_info& o = info[i];
            memset(&o, 0, sizeof(o));
            o.bi_threshold = sl.h_fscal[i].thr;
            o.bi_threshold_float = sl.h_fscal[i].thr_f;
            o.snr = sl.h_fscal[i].snr;
            o.sigma = sl.h_fscal[i].sigma;
        }
    }
    sl.busy = false;
    c->last_slot = c->oldest;
    c->oldest = (c->oldest + 1) % c->nslot;
    c->in_flight -= 1;
    return MDH_OK;
}

// ---- the same three steps without a host round trip (round 3): everything is QUEUED; the noise sums stay in device memory, the
// all-reduce and the all-to-all are enqueued by the host on this context's own streams (RCCL through mdh_stream_handle, or the
// single-process stand-ins below), and only mdh_band_collect / mdh_collect_batch wait.
static hipStream_t pick_stream(mdh_ctx* c, int which) {
    switch (which) {
        case MDH_STREAM_NOISE: return c->nstream;
        case MDH_STREAM_POST_LAST: return post_stream(c, c->pass_seq + 1);  // (pass_seq - 1) & 1
        case MDH_STREAM_POST_NEXT: return post_stream(c, c->pass_seq);
        default: return nullptr;
    }
}
extern "C" void* mdh_stream_handle(mdh_ctx* c, int which) { return c ? (void*)pick_stream(c, which) : nullptr; }

extern "C" int mdh_band_submit(mdh_ctx* c, const uint8_t* frames, int K, int on_device, int* n_jobs) {
    if (!c || !frames || !n_jobs) { set_err("mdh_band_submit: null argument"); return MDH_E_ARG; }
    if (c->role != MDH_ROLE_BAND) { set_err("mdh_band_submit: not a MDH_ROLE_BAND context"); return MDH_E_STATE; }
    if (c->band_slot) { set_err("mdh_band_submit: the previous pass still waits for its pixel stages"); return MDH_E_STATE; }
    CHECK_POISON(c, "mdh_band_submit");
    HIPCHK(hipSetDevice(c->device));
    Slot* sl = nullptr;
    int rc, nj = 0;
    if ((rc = slot_begin(c, K, sl)) != MDH_OK) return rc;
    if ((rc = stage_frames(c, *sl, frames, K, on_device)) != MDH_OK) return poison(c, rc);
    if ((rc = retire_frames(c, *sl, K, on_device)) != MDH_OK) return poison(c, rc);
    if ((rc = noise_prepare(c, *sl, K, &nj)) != MDH_OK) return poison(c, rc);
    c->band_slot = sl;
    c->band_thr_done = false;
    c->band_nj = nj;
    *n_jobs = nj;
    return MDH_OK;
}
extern "C" void* mdh_band_sums_device(mdh_ctx* c) { return (c && c->band_slot) ? (void*)c->band_slot->d_sums : nullptr; }

// The thresholds of the staged pass on their own, ahead of mdh_band_pixels_submit (which then skips this part).  For a process that
// drives SEVERAL bands on one device: the scalar chain is one workgroup of sixteen waves and 32 KB of LDS, and queued between two
// bands' window kernels -- each of which fills every CU -- it waits for a whole CU to drain (0.7 ms) with the next band's window
// kernel waiting behind it.  Queued for all bands BEFORE the first band's window kernel, the chains run while the device is quiet.
extern "C" int mdh_band_thresholds_submit(mdh_ctx* c, const int64_t* d_sums_total, double roi_px_total) {
    if (!c) { set_err("mdh_band_thresholds_submit: null context"); return MDH_E_ARG; }
    if (c->role != MDH_ROLE_BAND) { set_err("mdh_band_thresholds_submit: not a MDH_ROLE_BAND context"); return MDH_E_STATE; }
    if (!c->band_slot) { set_err("mdh_band_thresholds_submit: call mdh_band_submit first"); return MDH_E_STATE; }
    if (c->band_thr_done) { set_err("mdh_band_thresholds_submit: the staged pass has its thresholds already"); return MDH_E_STATE; }
    if (!(roi_px_total > 0)) { set_err("mdh_band_thresholds_submit: bad roi_px_total"); return MDH_E_ARG; }
    CHECK_POISON(c, "mdh_band_thresholds_submit");
    HIPCHK(hipSetDevice(c->device));
    Slot& sl = *c->band_slot;
    const int nj = c->band_nj;
    if (nj && d_sums_total && (const void*)d_sums_total != (const void*)c->d_sums)
        HIPCHK(hipMemcpyAsync(c->d_sums, d_sums_total, sizeof(long long) * 2 * nj, hipMemcpyDeviceToDevice, c->nstream));
    int rc;
    if ((rc = noise_finish(c, sl, sl.K, nj, roi_px_total)) != MDH_OK) return poison(c, rc);
    c->band_thr_done = true;
    return MDH_OK;
}

static int band_pixels_submit_impl(mdh_ctx* c, const int64_t* d_sums_total, double roi_px_total, int own_row0, int own_rows,
                                   int rows_stride, int nsplit, uint64_t* dst_rows_device, int flags, mdh_ctx* hough, int full_row0);
extern "C" int mdh_band_pixels_submit(mdh_ctx* c, const int64_t* d_sums_total, double roi_px_total, int own_row0, int own_rows,
                                      int rows_stride, int nsplit, uint64_t* dst_rows_device, int flags) {
    return band_pixels_submit_impl(c, d_sums_total, roi_px_total, own_row0, own_rows, rows_stride, nsplit, dst_rows_device, flags, nullptr, 0);
}
// A band of a process whose bands share ONE Hough context (same device): the owned rows go straight into the planes of the pass
// that context queues next, rows full_row0 ..., and their lit counts and bounding box into that pass's counters (k_export_rows_count)
// -- mdh_hough_planes_submit(hough, mdh_hough_input_planes(hough), K, 1, 1, H) then has nothing to assemble or count.  The first
// band of a pass zeroes the counters (on the Hough context's stream; every band's export waits for it).
extern "C" int mdh_band_pixels_submit_to(mdh_ctx* c, const int64_t* d_sums_total, double roi_px_total, int own_row0, int own_rows,
                                         mdh_ctx* hough, int full_row0) {
    if (!c || !hough) { set_err("mdh_band_pixels_submit_to: null argument"); return MDH_E_ARG; }
    if (hough->role != MDH_ROLE_HOUGH) { set_err("mdh_band_pixels_submit_to: the target is not a MDH_ROLE_HOUGH context"); return MDH_E_STATE; }
    if (hough->device != c->device) { set_err("mdh_band_pixels_submit_to: the two contexts are on different devices"); return MDH_E_ARG; }
    if (hough->WW != c->WW || hough->W != c->W) { set_err("mdh_band_pixels_submit_to: the two contexts differ in width"); return MDH_E_ARG; }
    if (hough->in_flight >= hough->nslot) { set_err("mdh_band_pixels_submit_to: the Hough context has no free pass slot"); return MDH_E_STATE; }
    if (full_row0 < 0 || own_rows < 1 || full_row0 + own_rows > hough->H) { set_err("mdh_band_pixels_submit_to: rows outside the whole frame"); return MDH_E_ARG; }
    if (c->band_slot && c->band_slot->K > hough->Kmax) { set_err("mdh_band_pixels_submit_to: the pass is larger than the Hough context's max_batch"); return MDH_E_ARG; }
    if (c->band_slot && hough->planes_begun && c->band_slot->K != hough->planes_K) { set_err("mdh_band_pixels_submit_to: the bands of one pass must export the same number of frames"); return MDH_E_ARG; }
    CHECK_POISON(hough, "mdh_band_pixels_submit_to");
    return band_pixels_submit_impl(c, d_sums_total, roi_px_total, own_row0, own_rows, hough->H, 1,
                                   (uint64_t*)hough->slot[hough->next_slot].d_dst, MDH_EXPORT_NOFILL, hough, full_row0);
}
static int band_pixels_submit_impl(mdh_ctx* c, const int64_t* d_sums_total, double roi_px_total, int own_row0, int own_rows,
                                   int rows_stride, int nsplit, uint64_t* dst_rows_device, int flags, mdh_ctx* hough, int full_row0) {
    if (!c || !dst_rows_device) { set_err("mdh_band_pixels_submit: null argument"); return MDH_E_ARG; }
    if (c->role != MDH_ROLE_BAND) { set_err("mdh_band_pixels_submit: not a MDH_ROLE_BAND context"); return MDH_E_STATE; }
    if (!c->band_slot) { set_err("mdh_band_pixels_submit: call mdh_band_submit first"); return MDH_E_STATE; }
    if (own_row0 < 0 || own_rows < 1 || own_row0 + own_rows > c->H || rows_stride < own_rows || nsplit < 1 || !(roi_px_total > 0)) {
        set_err("mdh_band_pixels_submit: bad owned rows / stride / split / roi_px_total");
        return MDH_E_ARG;
    }
    CHECK_POISON(c, "mdh_band_pixels_submit");
    HIPCHK(hipSetDevice(c->device));
    Slot& sl = *c->band_slot;
    const int K = sl.K, nj = c->band_nj;
    int rc;
    if (!c->band_thr_done) {  // (else: mdh_band_thresholds_submit has queued this part)
        // the global sums: already in place (in-place all-reduce on the noise stream), or somewhere else in device memory
        if (nj && d_sums_total && (const void*)d_sums_total != (const void*)c->d_sums)
            HIPCHK(hipMemcpyAsync(c->d_sums, d_sums_total, sizeof(long long) * 2 * nj, hipMemcpyDeviceToDevice, c->nstream));
        if ((rc = noise_finish(c, sl, K, nj, roi_px_total)) != MDH_OK) return poison(c, rc);
    }
    c->band_thr_done = false;
    HoughSet& hs = c->hset[c->pass_seq % c->nhset];
    hipStream_t post = post_stream(c, c->pass_seq);
    if ((rc = run_pixels(c, sl, K)) != MDH_OK) return poison(c, rc);
    const int Kp = (K + nsplit - 1) / nsplit;
    const size_t out_words = (size_t)nsplit * Kp * rows_stride * c->WW;
    u64* dev_out = (u64*)dst_rows_device;
    // (zero fill only where the export leaves holes: slots of frames that do not exist, rows a short last band does not own)
    if (!(flags & MDH_EXPORT_NOFILL) && ((size_t)nsplit * Kp != (size_t)K || own_rows != rows_stride))
        HIPCHK(hipMemsetAsync(dev_out, 0, out_words * sizeof(u64), post));
    if (hough) {
        Slot& hs_sl = hough->slot[hough->next_slot];
        if (!hough->planes_begun) {  // first band of the pass: the counters the bands accumulate into start at zero
            hipStream_t hpost = post_stream(hough, hough->pass_seq);
            if (!hough->ev_begin) HIPCHK(hipEventCreateWithFlags(&hough->ev_begin, hipEventDisableTiming));
            HIPCHK(hipMemsetAsync(hs_sl.d_rowcount, 0, sizeof(int) * ((size_t)hough->Kmax * hough->H + 5 * (size_t)hough->Kmax), hpost));
            HIPCHK(hipEventRecord(hough->ev_begin, hpost));
            hough->planes_begun = true;
            hough->planes_K = K;
        }
        HIPCHK(hipStreamWaitEvent(post, hough->ev_begin, 0));
        hipLaunchKernelGGL(k_export_rows_count, dim3((own_rows * c->WW + 255) / 256, K), dim3(256), 0, post, sl.d_dst, (u64*)hs_sl.d_dst, c->H,
                           c->WW, own_row0, own_rows, hough->H, c->W, full_row0, hs_sl.d_rowcount, hs_sl.d_litcount, hs_sl.d_bbox);
    } else
    hipLaunchKernelGGL(k_export_rows, dim3((own_rows * c->WW + 255) / 256, K), dim3(256), 0, post, sl.d_dst, dev_out, K, c->H, c->WW,
                       own_row0, own_rows, rows_stride, nsplit, Kp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(sl.h_fscal, sl.d_fscal, sizeof(FrameScalars) * K, hipMemcpyDeviceToHost, post));
    HIPCHK(hipMemcpyAsync(sl.h_off + K + 1, sl.dense ? (const void*)(sl.d_nzcount + 2) : (const void*)sl.d_dense, 2 * sizeof(unsigned),
                          hipMemcpyDeviceToHost, post));  // the format cue (see k_pack_lines)
    HIPCHK(hipEventRecord(sl.ev_done, post));
    const int si = (int)(&sl - c->slot);
    hs.last_slot = si;
    c->prev_slot = si;
    c->pass_seq += 1;
    slot_commit(c, sl);
    c->band_slot = nullptr;
    return MDH_OK;
}

extern "C" int mdh_band_collect(mdh_ctx* c, mdh_frame_info* info) {
    if (!c) { set_err("mdh_band_collect: null context"); return MDH_E_ARG; }
    if (c->role != MDH_ROLE_BAND) { set_err("mdh_band_collect: not a MDH_ROLE_BAND context"); return MDH_E_STATE; }
    if (c->in_flight <= 0) { set_err("mdh_band_collect: no pass in flight"); return MDH_E_STATE; }
    CHECK_POISON(c, "mdh_band_collect");
    HIPCHK(hipSetDevice(c->device));
    Slot& sl = c->slot[c->oldest];
    const int K = sl.K;
    {
        const int wrc = wait_event(c, sl.ev_done, "mdh_band_collect");
        if (wrc != MDH_OK) return wrc;
    }
    spans_collect(c, sl);
    format_cue(c, sl);
    if (info) {
        for (int i = 0; i < K; ++i) {
            mdh_frame_info& o = info[i];
            memset(&o, 0, sizeof(o));
            o.bi_threshold = sl.h_fscal[i].thr;
            o.bi_threshold_float = sl.h_fscal[i].thr_f;
            o.snr = sl.h_fscal[i].snr;
            o.sigma = sl.h_fscal[i].sigma;
        }
    }
    sl.busy = false;
    c->last_slot = c->oldest;
    c->oldest = (c->oldest + 1) % c->nslot;
    c->in_flight -= 1;
    return K;
}

// Stream-to-stream ordering between two contexts of one process (a band context's exported rows -> the Hough context that
// assembles them; an exchange enqueued on one stream -> its consumer on another): everything queued on `producer`'s stream so
// far precedes whatever is queued on `consumer`'s stream from now on.  No host wait.
extern "C" int mdh_chain(mdh_ctx* producer, int pwhich, mdh_ctx* consumer, int cwhich) {
    if (!producer || !consumer) { set_err("mdh_chain: null context"); return MDH_E_ARG; }
    hipStream_t ps = pick_stream(producer, pwhich), cs = pick_stream(consumer, cwhich);
    if (!ps || !cs) { set_err("mdh_chain: unknown stream designator"); return MDH_E_ARG; }
    if (producer->device != consumer->device) {  // (the event lives on one device; cross-GPU ordering is the collective's job)
        set_err("mdh_chain: the two contexts are on different devices");
        return MDH_E_ARG;
    }
    CHECK_POISON(producer, "mdh_chain");
    CHECK_POISON(consumer, "mdh_chain");
    HIPCHK(hipSetDevice(consumer->device));
    if (consumer->chain_ev.empty()) {
        consumer->chain_ev.resize(64, nullptr);
        for (auto& e : consumer->chain_ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    hipEvent_t ev = consumer->chain_ev[consumer->chain_next++ % consumer->chain_ev.size()];
    HIPCHK(hipEventRecord(ev, ps));
    HIPCHK(hipStreamWaitEvent(cs, ev, 0));
    return MDH_OK;
}

__global__ void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16);
extern "C" int mdh_device_copy_on(mdh_ctx* c, int which, void* dst, const void* src, uint64_t bytes) {
    if (!c || !dst || !src) { set_err("mdh_device_copy_on: null argument"); return MDH_E_ARG; }
    hipStream_t st = pick_stream(c, which);
    if (!st) { set_err("mdh_device_copy_on: unknown stream designator"); return MDH_E_ARG; }
    CHECK_POISON(c, "mdh_device_copy_on");
    HIPCHK(hipSetDevice(c->device));
    // a copy KERNEL, not hipMemcpyAsync: same-device copies of tens of megabytes go through an SDMA engine at ~65 GB/s
    // (8 bands x 8 pieces of 16.6 MB per pass: 16 ms), a grid of waves moves them at HBM speed
    if ((((uintptr_t)dst | (uintptr_t)src | bytes) & 15) == 0) {
        const size_t n16 = bytes / 16;
        const unsigned grid = (unsigned)std::min<size_t>((n16 + 255) / 256, 2048);
        hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, n16);
        HIPCHK(hipGetLastError());
    } else {
        HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
    }
    return MDH_OK;
}

// Single-process stand-in of the noise all-reduce: the staged passes of `n` band contexts on ONE device get the sum of their
// sums, on the device, ordered by events between the contexts' noise streams (what ncclAllReduce does across GPUs).
__global__ void __launch_bounds__(256) k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n16; e += (size_t)gridDim.x * blockDim.x) dst[e] = src[e];
}
struct SumPtrs { long long* p[MDH_MAX_LOCAL_BANDS]; };
__global__ void k_sum_bands(SumPtrs ptrs, int n, int count) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    long long t = 0;
    for (int i = 0; i < n; ++i) t += ptrs.p[i][j];
    for (int i = 0; i < n; ++i) ptrs.p[i][j] = t;
}
extern "C" int mdh_bands_allreduce_local(mdh_ctx** bands, int n) {
    if (!bands || n < 1 || n > MDH_MAX_LOCAL_BANDS) { set_err("mdh_bands_allreduce_local: 1..MDH_MAX_LOCAL_BANDS contexts"); return MDH_E_ARG; }
    SumPtrs ptrs{};
    for (int i = 0; i < n; ++i) {
        mdh_ctx* c = bands[i];
        if (!c || c->role != MDH_ROLE_BAND || !c->band_slot || c->device != bands[0]->device || c->band_nj != bands[0]->band_nj) {
            set_err("mdh_bands_allreduce_local: every context must be a band context of one device with a staged pass of the same frames");
            return MDH_E_STATE;
        }
        ptrs.p[i] = c->band_slot->d_sums;
    }
    const int count = 2 * bands[0]->band_nj;
    if (count == 0 || n == 1) return MDH_OK;
    HIPCHK(hipSetDevice(bands[0]->device));
    hipStream_t s0 = bands[0]->nstream;
    for (int i = 1; i < n; ++i) {
        HIPCHK(hipEventRecord(bands[i]->band_slot->ev_k1, bands[i]->nstream));
        HIPCHK(hipStreamWaitEvent(s0, bands[i]->band_slot->ev_k1, 0));
    }
    hipLaunchKernelGGL(k_sum_bands, dim3((count + 255) / 256), dim3(256), 0, s0, ptrs, n, count);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(bands[0]->band_slot->ev_k1, s0));
    for (int i = 1; i < n; ++i) HIPCHK(hipStreamWaitEvent(bands[i]->nstream, bands[0]->band_slot->ev_k1, 0));
    return MDH_OK;
}

// Several band contexts of ONE process on one device (row bands tiled on a single GPU): every context brings six streams of its
// own, and the runtime maps all of them onto a handful of hardware queues -- two unrelated streams that land on one queue wait
// for each other (rocprofv3 of 8 bands, round 4: band b + 1's window kernel started when band b's export had finished; nothing
// overlapped, 23 % of the run had a kernel on the GPU).  A band context that borrows the streams of a sibling queues its stages on
// the SAME main / noise / post streams: the bands' window kernels follow each other on one stream while their post stages run on
// another, and the process needs no more queues than one context.  Call right after mdh_create, before the first pass; the
// donor must outlive the borrower.
extern "C" int mdh_share_streams(mdh_ctx* c, mdh_ctx* donor) {
    if (!c || !donor || c == donor) { set_err("mdh_share_streams: bad argument"); return MDH_E_ARG; }
    if (c->device != donor->device) { set_err("mdh_share_streams: the two contexts are on different devices"); return MDH_E_ARG; }
    if (c->t != 0 || c->in_flight || c->pending || c->band_slot || c->pass_seq != 0) {
        set_err("mdh_share_streams: the context has already run a pass");
        return MDH_E_STATE;
    }
    if (c->streams_borrowed) { set_err("mdh_share_streams: the context borrows its streams already"); return MDH_E_STATE; }
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    hipStream_t own[] = {c->mstream[0], c->mstream[1], c->nstream, c->cstream, c->pstream[0], c->pstream[1]};
    for (hipStream_t q : own)
        if (q) hipStreamDestroy(q);
    c->mstream[0] = donor->mstream[0];
    c->mstream[1] = donor->mstream[1];
    c->nstream = donor->nstream;
    c->cstream = donor->cstream;
    c->pstream[0] = donor->pstream[0];
    c->pstream[1] = donor->pstream[1];
    c->stream = c->mstream[0];
    c->streams_borrowed = true;
    return MDH_OK;
}

extern "C" int mdh_band_skip(mdh_ctx* c) {
    if (!c) { set_err("mdh_band_skip: null context"); return MDH_E_ARG; }
    if (c->role != MDH_ROLE_BAND) { set_err("mdh_band_skip: not a MDH_ROLE_BAND context"); return MDH_E_STATE; }
    if (!c->band_slot) { set_err("mdh_band_skip: call mdh_band_noise first"); return MDH_E_STATE; }
    HIPCHK(hipSetDevice(c->device));
    Slot& sl = *c->band_slot;
    // the frames are in the window (mdh_band_noise staged and retired them); nothing else happens for this pass.
    // k_pass_prepare (queued by mdh_band_noise) has cleared the dst words of the slot's previous pass; the scalar chain that
    // would reset the slot's list counters does not run, so reset them here.
    if (sl.d_nzcount) HIPCHK(hipMemsetAsync(sl.d_nzcount, 0, 2 * sizeof(unsigned), c->nstream));
    HIPCHK(hipEventRecord(sl.ev_noise, c->nstream));
    HIPCHK(hipEventRecord(sl.ev_done, c->nstream));
    slot_commit(c, sl);
    c->band_slot = nullptr;
    HIPCHK(hipEventSynchronize(sl.ev_done));
    spans_collect(c, sl);
    sl.busy = false;
    sl.used = false;  // no dst words to clear, no binaries written
    c->oldest = (c->oldest + 1) % c->nslot;
    c->in_flight -= 1;
    return MDH_OK;
}

extern "C" int mdh_inject_sums(mdh_ctx* c, int64_t t, int64_t s1, int64_t s2) {
    if (!c) { set_err("mdh_inject_sums: null context"); return MDH_E_ARG; }
    if (c->role == MDH_ROLE_HOUGH) { set_err("mdh_inject_sums: a MDH_ROLE_HOUGH context has no noise chain"); return MDH_E_STATE; }
    if (t <= c->t) { set_err("mdh_inject_sums: the frame has already been submitted"); return MDH_E_ARG; }
    {   // only a frame whose sigma is due (Detector.py:82,87) ever looks its entry up: anything else would sit in the table for good
        const int64_t std_interval = (int64_t)c->cfg.interval * c->n;
        const bool due = (t > c->n && t % std_interval == 0) || (t > 1 && t <= c->n);
        if (!due) { set_err("mdh_inject_sums: no noise sigma is due for that frame (t > n and t % (interval * n) == 0, or 1 < t <= n)"); return MDH_E_ARG; }
    }
    for (auto& e : c->injected)
        if (e.first == t) { e.second = {s1, s2}; return MDH_OK; }
    if (c->injected.size() >= 64) { set_err("mdh_inject_sums: too many pending entries"); return MDH_E_STATE; }
    c->injected.push_back({t, {s1, s2}});
    return MDH_OK;
}

extern "C" int mdh_mute_hough_until(mdh_ctx* c, int64_t t) {
    if (!c) { set_err("mdh_mute_hough_until: null context"); return MDH_E_ARG; }
    c->mute_until = t;
    return MDH_OK;
}

extern "C" int mdh_seek(mdh_ctx* c, int64_t t0, int64_t ema_updates, double snr, double sigma, double thr_float, int thr) {
    if (!c) { set_err("mdh_seek: null context"); return MDH_E_ARG; }
    if (c->role == MDH_ROLE_HOUGH) { set_err("mdh_seek: a MDH_ROLE_HOUGH context has no stream position"); return MDH_E_STATE; }
    if (c->in_flight || c->pending || c->band_slot) { set_err("mdh_seek: collect / finish the passes in flight first"); return MDH_E_STATE; }
    if (t0 < 0 || ema_updates < 0) { set_err("mdh_seek: negative position"); return MDH_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    mail_cancel(c);
    HIPCHK(hipDeviceSynchronize());
    c->t = t0;
    c->injected.clear();
    c->mute_until = 0;
    c->seek_at = t0 > 0 ? t0 : -1;
    c->hist.assign(c->n - 1, c->d_zero);  // the window restarts empty: the next 2(n-1) frames are warm-up
    c->ema_t = ema_updates;
    c->ema_warm = c->n;  // the schedule (utils.py:360-368) is a function of the update count alone; it re-derives "ramp over"
    ScalarState st{};
    st.ema_v = snr;
    st.sigma = sigma;
    st.thr_f = thr_float;
    st.thr = thr;
    HIPCHK(hipMemcpy(c->d_state, &st, sizeof(st), hipMemcpyHostToDevice));
    // binaries of the previous position must not be taken for frames of the new one (the ring is tagged by frame number);
    // a context that has not run a pixel stage yet holds none (gigabytes of fills saved per time window)
    if (c->pass_seq > 0) {
        if (c->d_act) HIPCHK(hipMemset(c->d_act, 0, sizeof(ActWord) * (size_t)c->R * c->words));
        for (int si = 0; si < c->nslot; ++si)
            if (c->slot[si].d_bbits) HIPCHK(hipMemset(c->slot[si].d_bbits, 0, sizeof(ActWord) * (size_t)c->Kmax * c->words));
        for (int si = 0; si < c->nslot; ++si)
            if (c->slot[si].d_cplanes) HIPCHK(hipMemset(c->slot[si].d_cplanes, 0, sizeof(ActWord) * 2 * (size_t)c->Kmax * c->words));
    }
    for (int si = 0; si < c->nslot; ++si) {
        Slot& sl = c->slot[si];
        if (sl.used && sl.d_dst && sl.d_nzlist) {
            hipLaunchKernelGGL(k_sparse_clear, dim3(64), dim3(256), 0, c->nstream, sl.d_dst, sl.d_nzlist, sl.d_nzcount);
            sl.used = false;
        }
    }
    HIPCHK(hipStreamSynchronize(c->nstream));
    // hipMemset on device memory returns before the fill has run, and the library's streams are non-blocking ones that do not
    // wait for the null stream: without this, the passes submitted next race the fills above and lose tagged words to them
    // (seen with frames resident in HBM, where nothing slows the first pass down; host frames hid it behind their PCIe copy)
    HIPCHK(hipDeviceSynchronize());
    c->prev_slot = -1;
    c->last_slot = -1;
    return MDH_OK;
}

static int hough_planes_submit(mdh_ctx* c, const uint64_t* pieces, int K, int on_device, int nbands, int band_rows, const char* who) {
    if (!c || !pieces) { set_err(std::string(who) + ": null argument"); return MDH_E_ARG; }
    if (c->role != MDH_ROLE_HOUGH) { set_err(std::string(who) + ": not a MDH_ROLE_HOUGH context"); return MDH_E_STATE; }
    if (nbands < 1 || band_rows < 1 || (long long)nbands * band_rows < c->H) {
        set_err(std::string(who) + ": the bands do not cover the frame");
        return MDH_E_ARG;
    }
    CHECK_POISON(c, "mdh_hough_planes_submit");
    if (K < 1 || K > c->Kmax) { set_err(std::string(who) + ": pass size must be 1..max_batch"); return MDH_E_ARG; }
    if (c->in_flight >= c->nslot) { set_err(std::string(who) + ": the maximum number of passes is already in flight: collect one first"); return MDH_E_STATE; }
    // Every argument / state check comes BEFORE the slot is taken and before the "bands already counted into this slot" state is
    // consumed: an error return leaves the context exactly as it was (the caller may correct the call and repeat it).
    const bool counted = c->planes_begun;  // mdh_band_pixels_submit_to: the bands have filled the planes AND the counters of this slot
    const void* own_planes = (const void*)c->slot[c->next_slot].d_dst;  // (the slot slot_begin hands out next)
    // pieces == the slot's own plane buffer (mdh_hough_input_planes: the bands exported straight into it, one "band" of H rows):
    // nothing to copy, only the row / frame lit counts and the bounding boxes are taken
    const int inplace = (on_device && (const void*)pieces == own_planes && nbands == 1 && band_rows == c->H) ? 1 : 0;
    if (on_device && (const void*)pieces == own_planes && !inplace) { set_err(std::string(who) + ": the context's own plane buffer wants nbands = 1, band_rows = H"); return MDH_E_ARG; }
    if (counted && !inplace) { set_err(std::string(who) + ": bands exported into this pass with mdh_band_pixels_submit_to: submit the context's own planes"); return MDH_E_STATE; }
    if (counted && K != c->planes_K) {
        set_err(std::string(who) + ": the bands exported " + std::to_string(c->planes_K) + " frames into this pass, " + std::to_string(K) + " submitted");
        return MDH_E_ARG;
    }
    HIPCHK(hipSetDevice(c->device));
    Slot* slp = nullptr;
    int rc;
    if ((rc = slot_begin(c, K, slp)) != MDH_OK) return rc;
    Slot& sl = *slp;
    hipStream_t post = post_stream(c, c->pass_seq);
    const size_t in_words = (size_t)nbands * K * band_rows * c->WW;
    const u64* src = (const u64*)pieces;
    if (!on_device) {
        if ((rc = xchg_reserve(c, in_words)) != MDH_OK) return rc;
        HIPCHK(hipMemcpyAsync(c->d_xchg, pieces, in_words * sizeof(u64), hipMemcpyHostToDevice, post));
        src = c->d_xchg;
    }
    c->planes_begun = false;
    if (!counted) HIPCHK(hipMemsetAsync(sl.d_rowcount, 0, sizeof(int) * ((size_t)c->Kmax * c->H + 5 * (size_t)c->Kmax), post));
    if (!counted)
    hipLaunchKernelGGL(k_assemble_planes, dim3((unsigned)((c->words + 255) / 256), K), dim3(256), 0, post, src, sl.d_dst, sl.d_rowcount,
                       sl.d_litcount, sl.d_bbox, K, c->H, c->W, c->WW, band_rows, inplace);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(sl.ev_pix, post));
    // the staging buffer of host pieces is one per context: the next submit may refill it only after this pass has read it
    if (!on_device) HIPCHK(hipStreamSynchronize(post));
    if ((rc = run_hough(c, sl, K)) != MDH_OK) return rc;
    slot_commit(c, sl);
    return MDH_OK;
}

// The dst bit planes ([K][H][WW] u64, frame-major) the NEXT pass queued on a Hough context will read: band contexts of the same
// process may export their rows straight into them (mdh_band_pixels_submit with rows_stride = H, MDH_EXPORT_NOFILL) and hand the
// same pointer to mdh_hough_planes_submit (nbands = 1, band_rows = H), which then assembles in place.
extern "C" void* mdh_hough_input_planes(mdh_ctx* c) {
    if (!c || c->role != MDH_ROLE_HOUGH || c->in_flight >= c->nslot) return nullptr;
    return (void*)c->slot[c->next_slot].d_dst;
}

extern "C" int mdh_hough_planes(mdh_ctx* c, const uint64_t* pieces, int K, int on_device, int nbands, int band_rows,
                                mdh_frame_info* info, int32_t* raw_lines, int32_t* nms_lines, double* nonline) {
    const int rc = hough_planes_submit(c, pieces, K, on_device, nbands, band_rows, "mdh_hough_planes");
    if (rc != MDH_OK) return rc;
    return collect_pass(c, info, raw_lines, nms_lines, nonline, nullptr);
}

// The same, asynchronous: queues the pass (up to mdh_max_in_flight of them) and returns; mdh_collect_batch fetches the results of
// the oldest one.  Pieces in device memory must stay valid until k_assemble_planes of the pass has run, i.e. until the pass
// after the next one is submitted or this one is collected (two exchange buffers used alternately are enough).
extern "C" int mdh_hough_planes_submit(mdh_ctx* c, const uint64_t* pieces, int K, int on_device, int nbands, int band_rows) {
    return hough_planes_submit(c, pieces, K, on_device, nbands, band_rows, "mdh_hough_planes_submit");
}

extern "C" int mdh_get_binary(mdh_ctx* c, int index, int stage, uint8_t* out) {
    if (!c || !out) { set_err("mdh_get_binary: null argument"); return MDH_E_ARG; }
    if (c->last_slot < 0) { set_err("mdh_get_binary: no pass has been collected yet"); return MDH_E_STATE; }
    if (c->in_flight) { set_err("mdh_get_binary: collect the submitted passes first"); return MDH_E_STATE; }
    const Slot& sl = c->slot[c->last_slot];
    if (index < 0) index = sl.K - 1;
    if (index >= sl.K) { set_err("mdh_get_binary: index beyond the last pass"); return MDH_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    mail_cancel(c);
    const int total = (int)c->HW;
    if (stage == 0) {
        hipLaunchKernelGGL(k_expand_bits, dim3((total + 255) / 256), dim3(256), 0, c->stream, sl.d_dst + (size_t)index * c->words,
                           c->d_expand, c->H, c->W, c->WW);
    } else if (stage == 1 && c->cfg.detector == MDH_DET_M3) {
        const int64_t t = sl.t_base + 1 + index;
        hipLaunchKernelGGL(k_expand_act, dim3((total + 255) / 256), dim3(256), 0, c->stream,
                           c->d_act + (size_t)((t - 1) % c->R) * c->words, c->d_expand, c->H, c->W, c->WW, (u64)t);
    } else {
        set_err("mdh_get_binary: unknown stage");
        return MDH_E_ARG;
    }
    HIPCHK(hipMemcpyAsync(out, c->d_expand, c->HW, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MDH_OK;
}

extern "C" int mdh_debug_hough(mdh_ctx* c, long long* out, int K) {
    if (!c || !c->d_hdbg || K > c->Kmax) return MDH_E_ARG;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, c->d_hdbg, sizeof(long long) * 8 * K, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(out + (size_t)8 * K, c->d_hdbg + (size_t)c->Kmax * 8, sizeof(long long) * 8 * K, hipMemcpyDeviceToHost));
    return MDH_OK;
}

extern "C" void* mdh_device_alloc(mdh_ctx* c, uint64_t bytes) {
    if (!c) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || hipMalloc(&p, bytes) != hipSuccess) {
        set_err("mdh_device_alloc: hipMalloc failed");
        return nullptr;
    }
    return p;
}
extern "C" int mdh_device_free(mdh_ctx* c, void* p) {
    if (!c) return MDH_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipFree(p));
    return MDH_OK;
}
extern "C" int mdh_device_copy(mdh_ctx* c, void* dst, const void* src, uint64_t bytes) {
    if (!c || !dst || !src) return MDH_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    // done when the call returns, whatever the runtime makes of a device-to-device hipMemcpy: the consumers run on non-blocking streams
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->cstream));
    HIPCHK(hipStreamSynchronize(c->cstream));
    return MDH_OK;
}
extern "C" void* mdh_host_alloc(mdh_ctx* c, uint64_t bytes) {
    if (!c) return nullptr;
    void* p = nullptr;
    if (hipSetDevice(c->device) != hipSuccess || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        set_err("mdh_host_alloc: hipHostMalloc failed");
        return nullptr;
    }
    // (the per-frame protocol asks for a frame's device-visible address at every mdh_update: a pool the library handed out is
    // answered from this list instead of by the runtime)
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, p, 0) == hipSuccess && dp) {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        g_pools.push_back({(uintptr_t)p, (uintptr_t)bytes, (uintptr_t)dp, c->device, 0, c});
    } else (void)hipGetLastError();
    return p;
}
extern "C" int mdh_host_free(mdh_ctx* c, void* p) {
    if (!c) return MDH_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    mail_cancel(c);
    HIPCHK(hipDeviceSynchronize());
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (size_t k = 0; k < g_pools.size(); ++k)
            if (g_pools[k].base == (uintptr_t)p && g_pools[k].kind == 0) { g_pools.erase(g_pools.begin() + k); break; }
    }
    HIPCHK(hipHostFree(p));
    return MDH_OK;
}
// A loader that owns its frame buffers (a fixed ring in ordinary memory: multiprocessing.RawArray, a NumPy array it refills)
// registers them once; mdh_update then reads such a frame in place.  Without the call the library finds out by itself the third
// time an address comes by (mdh_set_auto_register).
extern "C" int mdh_host_register(mdh_ctx* c, void* p, uint64_t bytes) {
    if (!c || !p || bytes == 0) { set_err("mdh_host_register: bad argument"); return MDH_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (const HostPool& hp : g_pools)
            if (hp.device == c->device && (uintptr_t)p < hp.base + hp.bytes && hp.base < (uintptr_t)p + bytes) {
                set_err("mdh_host_register: the range overlaps memory that is registered already");
                return MDH_E_STATE;
            }
    }
    // memory the caller pinned itself (hipHostMalloc / hipHostRegister of its own) is announced the same way: it cannot be registered
    // a second time, and stays the caller's to release (round 6: the library no longer takes the runtime's word for what is pinned)
    int kind = 1;
    const hipError_t reg = hipHostRegister(p, bytes, hipHostRegisterMapped);
    if (reg != hipSuccess) {
        (void)hipGetLastError();
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess || at.type != hipMemoryTypeHost) {
            (void)hipGetLastError();
            set_err(std::string("mdh_host_register: hipHostRegister failed: ") + hipGetErrorString(reg));
            return MDH_E_HIP;
        }
        kind = 3;
    }
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, p, 0) != hipSuccess || !dp) {
        (void)hipGetLastError();
        if (kind == 1) hipHostUnregister(p);
        set_err("mdh_host_register: the registered range has no device-visible address");
        return MDH_E_HIP;
    }
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    g_pools.push_back({(uintptr_t)p, (uintptr_t)bytes, (uintptr_t)dp, c->device, kind, c});
    return MDH_OK;
}
extern "C" int mdh_host_unregister(mdh_ctx* c, void* p) {
    if (!c || !p) { set_err("mdh_host_unregister: null argument"); return MDH_E_ARG; }
    HIPCHK(hipSetDevice(c->device));
    mail_cancel(c);
    HIPCHK(hipDeviceSynchronize());
    bool found = false, ours = true;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (size_t k = 0; k < g_pools.size(); ++k)
            if (g_pools[k].base == (uintptr_t)p && g_pools[k].kind != 0) {
                if (g_pools[k].kind == 2 && g_pools[k].owner) g_pools[k].owner->auto_registered -= 1;
                ours = g_pools[k].kind != 3;
                g_pools.erase(g_pools.begin() + k);
                found = true;
                break;
            }
    }
    if (!found) { set_err("mdh_host_unregister: not a range registered through this library"); return MDH_E_ARG; }
    if (ours) HIPCHK(hipHostUnregister(p));
    return MDH_OK;
}
extern "C" int mdh_set_auto_register(mdh_ctx* c, int on) {
    if (!c) { set_err("mdh_set_auto_register: null context"); return MDH_E_ARG; }
    c->auto_register_on = on != 0;
    return MDH_OK;
}
extern "C" int mdh_update_stats(const mdh_ctx* c, int64_t* direct_frames, int64_t* copied_frames, int* auto_registered) {
    if (!c) return MDH_E_ARG;
    if (direct_frames) *direct_frames = c->direct_frames;
    if (copied_frames) *copied_frames = c->copied_frames;
    if (auto_registered) *auto_registered = c->auto_registered;
    return MDH_OK;
}
extern "C" int mdh_device_upload(mdh_ctx* c, void* dst, const void* src, uint64_t bytes) {
    if (!c || !dst || !src) return MDH_E_ARG;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return MDH_OK;
}

// ------------------------------------------------------------------------------------------------ preprocessing
struct mdh_pre {
    int device = 0;
    PreGeom g{};
    hipStream_t stream = nullptr;
    int *d_xi = nullptr, *d_yi = nullptr;
    short *d_xw = nullptr, *d_yw = nullptr;
    uint8_t *d_mask = nullptr, *d_src = nullptr, *d_out = nullptr;
    size_t src_cap = 0, out_cap = 0;
    int span_cap = 0;          // k_preprocess_rows: bytes of a source row one workgroup stages (0: the layout does not allow that kernel)
    uint64_t rows_touched = 0; // source rows the resize reads per source frame (the kernel's compulsory traffic: mdh_pre_traffic_model)
    // Frames handed over in HOST memory travel through two pinned buffers of the library's own (CPU copy of chunk k + 1 beside the
    // DMA of chunk k): the caller's pages are never pinned by the runtime on the fly.  Round 5: twice in ten sessions of the whole
    // GPU suite the process was aborted inside mdh_pre_run on the one test case whose pageable source exceeds a megabyte (6 MB,
    // on a stream created a moment before) -- not reproduced in isolation (600 iterations) nor under MALLOC_CHECK_; this path does
    // not depend on how the runtime moves pageable memory.  MDH_PRE_STAGE=0: hipMemcpyAsync straight from the caller's memory (A/B).
    uint8_t* h_stage[2] = {nullptr, nullptr};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
};
static const size_t PRE_STAGE_BYTES = 8u << 20;

// OpenCV resize(INTER_LINEAR) coefficient tables: fx = (float)((d+0.5)*scale-0.5), clamp, weights cvRound(w*2048)
static void linear_table(int dst_n, int src_n, std::vector<int>& idx, std::vector<short>& w) {
    idx.resize(dst_n);
    w.resize(2 * (size_t)dst_n);
    const double scale = 1.0 / ((double)dst_n / src_n);  // resize.cpp: scale = 1. / inv_scale
    for (int d = 0; d < dst_n; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= (float)s;
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= src_n - 1) { s = src_n - 1; f = 0.f; }
        idx[d] = s;
        w[2 * d] = (short)lrintf((1.f - f) * 2048.f);
        w[2 * d + 1] = (short)lrintf(f * 2048.f);
    }
}

extern "C" void mdh_pre_destroy(mdh_pre* p) {
    if (!p) return;
    hipSetDevice(p->device);
    if (p->stream) hipStreamSynchronize(p->stream);
    void* dev[] = {p->d_xi, p->d_yi, p->d_xw, p->d_yw, p->d_mask, p->d_src, p->d_out};
    for (void* q : dev)
        if (q) hipFree(q);
    for (int k = 0; k < 2; ++k) {
        if (p->h_stage[k]) hipHostFree(p->h_stage[k]);
        if (p->ev_stage[k]) hipEventDestroy(p->ev_stage[k]);
    }
    if (p->stream) hipStreamDestroy(p->stream);
    delete p;
}

static int pre_init(mdh_pre* p, const uint8_t* mask01) {
    HIPCHK(hipSetDevice(p->device));
    HIPCHK(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    std::vector<int> xi, yi;
    std::vector<short> xw, yw;
    linear_table(p->g.dw, p->g.sw, xi, xw);
    linear_table(p->g.dh, p->g.sh, yi, yw);
    {   // source rows the vertical taps touch (distinct values among yi[dy] and yi[dy] + 1)
        std::vector<char> used(p->g.sh, 0);
        for (int dy = 0; dy < p->g.dh; ++dy) { used[yi[dy]] = 1; used[std::min(yi[dy] + 1, p->g.sh - 1)] = 1; }
        for (char u : used) p->rows_touched += u;
    }
    if (p->g.sw % 16 == 0 && !(getenv("MDH_PRE_ROWS") && atoi(getenv("MDH_PRE_ROWS")) == 0)) {  // (MDH_PRE_ROWS=0: the one-thread-per-pixel kernel, A/B and parity)
        int cap = 0;
        for (int dx0 = 0; dx0 < p->g.dw; dx0 += PRE_SEG) {
            const int cnt = std::min(PRE_SEG, p->g.dw - dx0);
            const int c_first = xi[dx0], c_last = std::min(xi[dx0 + cnt - 1] + 1, p->g.sw - 1);
            const int byte0 = (c_first * 3) & ~15;
            cap = std::max(cap, (((c_last + 1) * 3 - byte0) + 15) & ~15);
        }
        if (2 * PRE_ROWS * cap <= 64 * 1024) p->span_cap = cap;
    }
    int rc;
    if ((rc = dev_alloc(p->d_xi, xi.size(), false)) || (rc = dev_alloc(p->d_yi, yi.size(), false)) ||
        (rc = dev_alloc(p->d_xw, xw.size(), false)) || (rc = dev_alloc(p->d_yw, yw.size(), false)) ||
        (rc = dev_alloc(p->d_out, (size_t)p->g.dh * p->g.dw, false)))
        return rc;
    p->out_cap = (size_t)p->g.dh * p->g.dw;
    HIPCHK(hipMemcpy(p->d_xi, xi.data(), xi.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->d_yi, yi.data(), yi.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->d_xw, xw.data(), xw.size() * sizeof(short), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(p->d_yw, yw.data(), yw.size() * sizeof(short), hipMemcpyHostToDevice));
    if (mask01) {
        if ((rc = dev_alloc(p->d_mask, (size_t)p->g.dh * p->g.dw, false))) return rc;
        HIPCHK(hipMemcpy(p->d_mask, mask01, (size_t)p->g.dh * p->g.dw, hipMemcpyHostToDevice));
    }
    return MDH_OK;
}

extern "C" mdh_pre* mdh_pre_create(int device, int src_h, int src_w, int dst_h, int dst_w, const uint8_t* mask01) {
    if (src_h <= 0 || src_w <= 0 || dst_h <= 0 || dst_w <= 0) {
        set_err("mdh_pre_create: bad geometry");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_err("mdh_pre_create: no such HIP device (there is no CPU fallback)");
        return nullptr;
    }
    mdh_pre* p = new mdh_pre();
    p->device = device;
    p->g = PreGeom{src_h, src_w, dst_h, dst_w};
    if (pre_init(p, mask01) != MDH_OK) {
        std::string keep = g_err;
        mdh_pre_destroy(p);
        set_err(keep);
        return nullptr;
    }
    return p;
}

extern "C" int mdh_pre_run_batch(mdh_pre* p, const uint8_t* bgr, int n_out, int exp_frame, int src_on_device, uint8_t* out,
                                 int out_on_device);
extern "C" int mdh_pre_run(mdh_pre* p, const uint8_t* bgr, int exp_frame, int src_on_device, uint8_t* out, int out_on_device) {
    return mdh_pre_run_batch(p, bgr, 1, exp_frame, src_on_device, out, out_on_device);
}

extern "C" int mdh_pre_run_batch(mdh_pre* p, const uint8_t* bgr, int n_out, int exp_frame, int src_on_device, uint8_t* out,
                                 int out_on_device) {
    if (!p || !bgr || !out || exp_frame < 1 || n_out < 1) { set_err("mdh_pre_run_batch: bad argument"); return MDH_E_ARG; }
    HIPCHK(hipSetDevice(p->device));
    const size_t frame = (size_t)p->g.sh * p->g.sw * 3, need = frame * exp_frame * n_out;
    const uint8_t* src = bgr;
    if (!src_on_device) {
        if (p->src_cap < need) {
            if (p->d_src) HIPCHK(hipFree(p->d_src));
            p->d_src = nullptr;
            HIPCHK(hipMalloc((void**)&p->d_src, need));
            p->src_cap = need;
        }
        static const bool stage = !(getenv("MDH_PRE_STAGE") && atoi(getenv("MDH_PRE_STAGE")) == 0);
        // Pinned memory is copied as it is -- but only memory THIS LIBRARY pinned (mdh_host_alloc) or registered at the caller's explicit
        // request (mdh_host_register) counts.  Up to round 5 the runtime was asked (hipPointerGetAttributes), and the runtime also
        // answers "host memory" for a range somebody registered once and freed without unregistering: a fresh array the allocator
        // places on such a range then took the direct copy through the stale mapping -- "Memory access fault by GPU", the process
        // aborted inside this call, staging or not (tools/stress_pre_run.py --mode stale reproduces it on round 5's library:
        // gpurun_out/r06_09).  Everything else goes through the staging buffers, where only the CPU reads it.
        bool pinned = false;
        {
            std::lock_guard<std::mutex> lock(g_pool_mutex);
            for (const HostPool& hp : g_pools)
                if (hp.kind != 2 && (uintptr_t)bgr >= hp.base && (uintptr_t)bgr + need <= hp.base + hp.bytes) { pinned = true; break; }
        }
        if (!stage || pinned) {
            HIPCHK(hipMemcpyAsync(p->d_src, bgr, need, hipMemcpyHostToDevice, p->stream));
        } else {
            for (int k = 0; k < 2; ++k) {
                if (!p->h_stage[k]) HIPCHK(hipHostMalloc((void**)&p->h_stage[k], PRE_STAGE_BYTES, hipHostMallocDefault));
                if (!p->ev_stage[k]) HIPCHK(hipEventCreateWithFlags(&p->ev_stage[k], hipEventDisableTiming));
            }
            int k = 0;
            for (size_t off = 0; off < need; off += PRE_STAGE_BYTES, k ^= 1) {
                const size_t len = std::min(PRE_STAGE_BYTES, need - off);
                if (off >= 2 * PRE_STAGE_BYTES) HIPCHK(hipEventSynchronize(p->ev_stage[k]));  // the DMA that last read this buffer
                memcpy(p->h_stage[k], bgr + off, len);
                HIPCHK(hipMemcpyAsync(p->d_src + off, p->h_stage[k], len, hipMemcpyHostToDevice, p->stream));
                HIPCHK(hipEventRecord(p->ev_stage[k], p->stream));
            }
        }
        src = p->d_src;
    }
    const int px = p->g.dh * p->g.dw;
    if (!out_on_device && p->out_cap < (size_t)px * n_out) {
        if (p->d_out) HIPCHK(hipFree(p->d_out));
        p->d_out = nullptr;
        HIPCHK(hipMalloc((void**)&p->d_out, (size_t)px * n_out));
        p->out_cap = (size_t)px * n_out;
    }
    uint8_t* dst = out_on_device ? out : p->d_out;
    // (MDH_PRE_PREFETCH=0: the kernel without the register prefetch of the next stage, as up to the first half of round 6)
    const bool pre_pf = !(getenv("MDH_PRE_PREFETCH") && atoi(getenv("MDH_PRE_PREFETCH")) == 0);  // (read per call: tests switch it inside one process)
    const bool rows_ok = p->span_cap > 0 && ((uintptr_t)src & 15) == 0 && n_out <= 65535 && p->g.dh <= 65535;
    const dim3 rows_grid((p->g.dw + PRE_SEG - 1) / PRE_SEG, (p->g.dh + PRE_ROWS - 1) / PRE_ROWS, (n_out + PRE_ZB - 1) / PRE_ZB);
    if (rows_ok && pre_pf && p->span_cap <= PRE_SEG * 16)
        hipLaunchKernelGGL(k_preprocess_rows_pf<1>, rows_grid, dim3(PRE_SEG), 2 * PRE_ROWS * (size_t)p->span_cap, p->stream,
                           src, exp_frame, p->g, p->d_xi, p->d_xw, p->d_yi, p->d_yw, p->d_mask, dst, p->span_cap, n_out);
    else if (rows_ok && pre_pf && p->span_cap <= 2 * PRE_SEG * 16)
        hipLaunchKernelGGL(k_preprocess_rows_pf<2>, rows_grid, dim3(PRE_SEG), 2 * PRE_ROWS * (size_t)p->span_cap, p->stream,
                           src, exp_frame, p->g, p->d_xi, p->d_xw, p->d_yi, p->d_yw, p->d_mask, dst, p->span_cap, n_out);
    else if (rows_ok)
        hipLaunchKernelGGL(k_preprocess_rows, dim3((p->g.dw + PRE_SEG - 1) / PRE_SEG, (p->g.dh + PRE_ROWS - 1) / PRE_ROWS, (n_out + PRE_ZB - 1) / PRE_ZB), dim3(PRE_SEG),
                           2 * PRE_ROWS * (size_t)p->span_cap, p->stream,
                           src, exp_frame, p->g, p->d_xi, p->d_xw, p->d_yi, p->d_yw, p->d_mask, dst, p->span_cap, n_out);
    else
    hipLaunchKernelGGL(k_preprocess, dim3((px + 255) / 256, n_out), dim3(256), 0, p->stream, src, exp_frame, p->g, p->d_xi, p->d_xw,
                       p->d_yi, p->d_yw, p->d_mask, dst);
    HIPCHK(hipGetLastError());
    if (!out_on_device) HIPCHK(hipMemcpyAsync(out, p->d_out, (size_t)px * n_out, hipMemcpyDeviceToHost, p->stream));
    HIPCHK(hipStreamSynchronize(p->stream));
    return MDH_OK;
}

// Bytes one launch of the preprocessing kernel must move per OUTPUT frame: the source rows the resize's vertical taps touch (whole
// rows: the horizontal taps of a reduction use bytes of every cache line of such a row), exp_frame source frames each, + the
// mask + the output.  (bench.py prices the kernel's roofline on this, next to the PMC-measured traffic.)
extern "C" uint64_t mdh_pre_traffic_model(const mdh_pre* p, int exp_frame) {
    if (!p || exp_frame < 1) return 0;
    const uint64_t out = (uint64_t)p->g.dh * p->g.dw;
    return p->rows_touched * (uint64_t)p->g.sw * 3 * (uint64_t)exp_frame + out + (p->d_mask ? out : 0);
}

extern "C" int mdh_max_stack(int device, const uint8_t* frames, int K, uint64_t frame_bytes, int on_device, uint8_t* out,
                             int out_on_device) {
    if (!frames || !out || K < 1 || frame_bytes == 0) { set_err("mdh_max_stack: bad argument"); return MDH_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_err("mdh_max_stack: no such HIP device (there is no CPU fallback)");
        return MDH_E_HIP;
    }
    HIPCHK(hipSetDevice(device));
    uint8_t *d_in = nullptr, *d_out = nullptr;
    const uint8_t* src = frames;
    if (!on_device) {
        HIPCHK(hipMalloc((void**)&d_in, (size_t)K * frame_bytes));
        HIPCHK(hipMemcpy(d_in, frames, (size_t)K * frame_bytes, hipMemcpyHostToDevice));
        src = d_in;
    }
    uint8_t* dst = out;
    if (!out_on_device) {
        HIPCHK(hipMalloc((void**)&d_out, frame_bytes));
        dst = d_out;
    }
    const size_t threads = std::max<size_t>(frame_bytes / 16, 16);
    hipLaunchKernelGGL(k_max_stack, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, 0, src, K, (size_t)frame_bytes, dst);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    if (!out_on_device) HIPCHK(hipMemcpy(out, d_out, frame_bytes, hipMemcpyDeviceToHost));
    if (d_in) hipFree(d_in);
    if (d_out) hipFree(d_out);
    return MDH_OK;
}

extern "C" int mdh_frame_sums(int device, const uint8_t* frames, int K, uint64_t frame_bytes, int on_device, uint64_t* sums) {
    if (!frames || !sums || K < 1 || frame_bytes == 0) { set_err("mdh_frame_sums: bad argument"); return MDH_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_err("mdh_frame_sums: no such HIP device (there is no CPU fallback)");
        return MDH_E_HIP;
    }
    HIPCHK(hipSetDevice(device));
    uint8_t* d_in = nullptr;
    unsigned long long* d_sums = nullptr;
    const uint8_t* src = frames;
    if (!on_device) {
        HIPCHK(hipMalloc((void**)&d_in, (size_t)K * frame_bytes));
        HIPCHK(hipMemcpy(d_in, frames, (size_t)K * frame_bytes, hipMemcpyHostToDevice));
        src = d_in;
    }
    if (((uintptr_t)src | frame_bytes) & 15) {  // the vector loop wants 16-byte aligned frames
        if (d_in) hipFree(d_in);
        set_err("mdh_frame_sums: frames must be 16-byte aligned and a multiple of 16 bytes long");
        return MDH_E_ARG;
    }
    HIPCHK(hipMalloc((void**)&d_sums, sizeof(unsigned long long) * K));
    HIPCHK(hipMemset(d_sums, 0, sizeof(unsigned long long) * K));
    const unsigned per_frame = (unsigned)std::min<size_t>(std::max<size_t>(frame_bytes / 16 / 256 / 4, 1), 64);
    hipLaunchKernelGGL(k_frame_sums, dim3(per_frame, K), dim3(256), 0, 0, src, (size_t)frame_bytes, d_sums);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(sums, d_sums, sizeof(unsigned long long) * K, hipMemcpyDeviceToHost));
    if (d_in) hipFree(d_in);
    hipFree(d_sums);
    return MDH_OK;
}


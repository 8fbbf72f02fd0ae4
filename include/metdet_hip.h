/* metdet_hip.h -- C-ABI of libmetdet_hip.so: the MI355X (gfx950) implementation of MetDetPy's
 * per-frame line-detector hot path.
 *
 * The reference is pure Python and has no FFI for this path; the entry points below are sized to the
 * plugin contract the reference's main loop uses (MetDetPy.py:137-142, 197-198):
 *
 *     detector = DetectorCls(window_sec, fps, mask, num_cls, cfg, logger)   Detector.py:186-220
 *     detector.update(frame)                                                Detector.py:225-229
 *     lines, cls_pred = detector.detect()                                   Detector.py:324-392 / 257-296
 *
 * and are what a ctypes stub inside the reference's `LineDetector` subclasses would bind (see
 * INTEGRATION.md).  Plain pointers and sizes only; no exceptions cross the boundary.  Every function
 * that returns int returns 0 on success and a negative MDH_E_* code on failure, with a text in
 * mdh_last_error().  One mdh_ctx is used by one host thread at a time.  The library owns all device
 * memory; host buffers handed in are only read during the call.
 */
#ifndef METDET_HIP_H
#define METDET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDH_MAX_LINES 512        /* raw Hough segments stored per frame (reference discards frames with > 500, Detector.py:30) */
#define MDH_MAX_WINDOW 255       /* n = int(window_sec * fps) must be 1..255 */
#define MDH_NUM_ANGLE 180        /* HoughLinesP(theta = pi/180) */

#define MDH_OK 0
#define MDH_E_ARG (-1)           /* bad argument */
#define MDH_E_HIP (-2)           /* a HIP runtime call failed */
#define MDH_E_STATE (-3)         /* call sequence error (detect without update, ...) */
#define MDH_E_NOMEM (-4)
#define MDH_E_TIMEOUT (-5)       /* a collect gave up waiting (mdh_set_wait_timeout_ms); the context is poisoned */

#define MDH_DET_M3 0             /* M3Detector.detect       (Detector.py:324-392) */
#define MDH_DET_CLASSIC 1        /* ClassicDetector.detect  (Detector.py:257-296) */

/* mdh_cfg.role: what a context runs (row-band tiling of one stream across GPUs, see the end of this file) */
#define MDH_ROLE_FULL 0          /* the whole path (default) */
#define MDH_ROLE_BAND 1          /* update() + the pixel stages of detect() on one band of rows (plus halo) */
#define MDH_ROLE_HOUGH 2         /* detect() from the binary image onwards, on bit planes assembled from the bands */
#define MDH_BAND_HALO 4          /* rows a band needs from each neighbour: median 1 + dilate 1 + erode 1 + dynamic-mask erode 1 */

#define MDH_SENS_LOW 0
#define MDH_SENS_NORMAL 1
#define MDH_SENS_HIGH 2

/* cfg.detector.cfg of the reference (metstruct.py:357-382) plus what LineDetector.__init__ derives on the host. */
/* ABI guard.  Both configuration structs start with `struct_size` = sizeof(the struct) as the CALLER's header declares it;
 * mdh_create / mdh_collector_create refuse (NULL, text in mdh_last_error) a struct whose size differs from the library's, so a
 * host compiled against an older header -- or a stale hand-written binding -- fails loudly instead of having fields read from
 * beyond its struct.  mdh_abi_version() is bumped whenever a struct or a prototype changes (2: struct_size fields, round 3;
 * 3: mdh_host_register / mdh_host_unregister / mdh_set_auto_register / mdh_update_stats, round 4 -- and, added later in round 4
 *    without a bump: MDH_E_TIMEOUT, mdh_set_wait_timeout_ms, mdh_share_streams, mdh_band_thresholds_submit,
 *    mdh_band_pixels_submit_to, mdh_device_reset, mdh_pre_traffic_model, mdh_set_device;
 * 4: round 5 -- everything listed under 3 is guaranteed present; auto-registration is OFF by default (mdh_set_auto_register);
 *    mdh_device_count / mdh_stream_create / mdh_stream_destroy / mdh_stream_sync / mdh_dev_malloc / mdh_dev_free / mdh_dev_copy
 *    (context-free helpers for a host's own collectives); mdh_hough_planes_submit validates before it takes a slot). */
#define MDH_ABI_VERSION 4
int mdh_abi_version(void);
/* The library runs a pass on up to ten HIP streams; the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4),
 * read once when the runtime initialises.  The library never modifies the process environment: export GPU_MAX_HW_QUEUES=12
 * before the first HIP call for full overlap.  Returns the value the environment holds (0 = unset).  Results never depend on it. */
int mdh_hw_queues_hint(void);
/* hipSetDevice(device) for hosts without a HIP binding of their own (the RCCL communicator binds to the current device). */
int mdh_set_device(int device);
/* Context-free device helpers, for a host that enqueues collectives of its own (metdetpy_amd/dist.py: the final detection gather
 * is an ncclAllGather on a private stream over a few bytes of device memory).  No mdh_ctx is involved. */
int mdh_device_count(void);                                  /* HIP devices visible to this process (0 when there is none) */
void* mdh_stream_create(int device);                         /* hipStream_t (non-blocking), NULL on failure */
int mdh_stream_destroy(int device, void* stream);
int mdh_stream_sync(int device, void* stream, double timeout_ms);   /* timeout_ms <= 0: for ever; else MDH_E_TIMEOUT when it expires */
void* mdh_dev_malloc(int device, uint64_t bytes);
int mdh_dev_free(int device, void* ptr);
int mdh_dev_copy(int device, void* dst, const void* src, uint64_t bytes, int to_device, void* stream);  /* queued on stream, then waited for */
/* hipDeviceReset(device): releases everything this process holds on the device (every mdh_ctx / mdh_pre on it must have been
 * destroyed first).  For a host that is done with the GPU but stays alive while other processes use it. */
int mdh_device_reset(int device);

typedef struct mdh_cfg {
    uint32_t struct_size;        /* sizeof(mdh_cfg): checked first by mdh_create */
    int32_t detector;            /* MDH_DET_* */
    int32_t adaptive_bi_thre;    /* cfg.binary.adaptive_bi_thre */
    int32_t init_value;          /* cfg.binary.init_value (used when not adaptive, Detector.py:208) */
    int32_t sensitivity;         /* MDH_SENS_* : T = a*snr^2+b (Detector.py:177-183) */
    int32_t interval;            /* cfg.binary.interval (Detector.py:61-63) */
    int32_t roi[4];              /* SNR_SW.std_roi = (row0, col0, row1, col1) chosen by select_subarea (Detector.py:93-122) */
    int32_t hough_threshold;     /* cfg.hough_line.threshold */
    int32_t hough_min_len;       /* cfg.hough_line.min_len */
    int32_t hough_max_gap;       /* cfg.hough_line.max_gap */
    int32_t dy_mask;             /* cfg.dynamic.dy_mask */
    int32_t max_batch;           /* largest K accepted by one internal pass of mdh_process_batch (0 -> 64) */
    int32_t role;                /* MDH_ROLE_* */
    double mask_area;            /* float(np.sum(mask)) (Detector.py:191) */
    int32_t hough_mode;          /* MDH_HOUGH_* : 0 = exact replay of cv2.HoughLinesP (the parity path), 1 = full accumulator */
    int32_t reserved;            /* 0 */
} mdh_cfg;

/* hough_mode.  MDH_HOUGH_EXACT replays OpenCV's progressive probabilistic transform (same random visiting order, same
 * segments in the same order: Detector.py:347-353).  MDH_HOUGH_ACCUMULATOR is a different, order-free algorithm on the same
 * kernels' data structures: every lit pixel votes every angle (LDS atomics), then the strongest accumulator cell is taken
 * while it holds >= threshold votes and a lit pixel on its line seeds the same walk / clear / take-back procedure.  Its
 * segments are NOT OpenCV's; they agree with them under SURVEY 8d's tolerance rule (rho within 1 px and theta within 0.5 deg
 * for isolated lines; identical collector records on streak streams: tests/test_gpu_hough_accumulator.py).  Frames with more
 * than 8192 lit pixels are replayed exactly in both modes. */
#define MDH_HOUGH_EXACT 0
#define MDH_HOUGH_ACCUMULATOR 1

/* Per-frame scalars: what M3Detector exposes as attributes after detect() (Detector.py:342-392, 394-448). */
typedef struct mdh_frame_info {
    int32_t n_raw;               /* lines_num: segments HoughLinesP returned (may exceed MDH_MAX_LINES; only that many are stored) */
    int32_t n_nms;               /* filtered_line_num: segments after lineset_nms (0 when n_raw == 0 or n_raw > 500) */
    int32_t lit_count;           /* count(dst == 255) */
    int32_t bi_threshold;        /* bi_threshold in force for this frame */
    double bi_threshold_float;
    double snr;                  /* stack.snr (noise EMA) after this frame's update */
    double sigma;                /* last noise sigma fed to the EMA (0 before the first) */
    double dst_sum;              /* 100 * lit_count / mask_area */
    double hough_gap;            /* maxLineGap handed to HoughLinesP (before cvRound) */
} mdh_frame_info;

typedef struct mdh_ctx mdh_ctx;

/* LineDetector.__init__ (Detector.py:186-220).  H, W: runtime frame size; n_window = int(window_sec*fps).
 * Returns NULL on failure (see mdh_last_error). */
mdh_ctx* mdh_create(int device, int H, int W, int n_window, const mdh_cfg* cfg);
void mdh_destroy(mdh_ctx* ctx);
const char* mdh_last_error(void);

/* LineDetector.update(frame) (Detector.py:225-229): copies the H*W u8 host frame into the window and
 * refreshes the noise estimate / binary threshold.  Must alternate with mdh_detect. */
int mdh_update(mdh_ctx* ctx, const uint8_t* frame);

/* {M3,Classic}Detector.detect() (Detector.py:324-392 / 257-296) for the frame given to the last mdh_update.
 * raw_lines: MDH_MAX_LINES x 4 int32 [x1,y1,x2,y2] in HoughLinesP discovery order; nms_lines / nonline:
 * output of lineset_nms (utils.py:780-839), nonline = "non-line probability".  Any output may be NULL. */
int mdh_detect(mdh_ctx* ctx, mdh_frame_info* info, int32_t* raw_lines, int32_t* nms_lines, double* nonline);

/* Throughput path: K consecutive update()+detect() pairs in one call, identical results.
 * frames: K*H*W u8, host memory (on_device = 0) or device memory of ctx's device (on_device = 1).
 * info: K entries.  raw_lines / nms_lines: K*MDH_MAX_LINES*4 int32, nonline: K*MDH_MAX_LINES doubles
 * (each may be NULL).  K may exceed cfg.max_batch; it is processed in passes of at most max_batch frames. */
int mdh_process_batch(mdh_ctx* ctx, const uint8_t* frames, int K, int on_device, mdh_frame_info* info,
                      int32_t* raw_lines, int32_t* nms_lines, double* nonline);

/* Asynchronous form of mdh_process_batch for K <= cfg.max_batch: mdh_submit_batch queues the pass and returns at
 * once; mdh_collect_batch waits for the OLDEST queued pass and returns its frame count (> 0) or a negative error.
 * At most mdh_max_in_flight(ctx) passes (6 by default) may be in flight; the window kernel of a later pass overlaps the
 * morphology / Hough stages of the earlier ones.
 * `frames` must stay valid (and, for device memory, unmodified) until its pass has been collected. */
int mdh_submit_batch(mdh_ctx* ctx, const uint8_t* frames, int K, int on_device);
int mdh_collect_batch(mdh_ctx* ctx, mdh_frame_info* info, int32_t* raw_lines, int32_t* nms_lines, double* nonline);
int mdh_in_flight(const mdh_ctx* ctx);
int mdh_max_in_flight(const mdh_ctx* ctx);
/* Diagnostics of the window kernel's output format (sparse + tagged words on a quiet sky; plain dense words when the binaries carry
 * noise, e.g. 8K at m3det_high: chosen per pass from what the last collected pass reported (half of its frames dense -> dense format; below 1.5 % of the words set -> back);
 * results are identical either way).  MDH_DENSE=0|1 in the environment pins the format. */
int64_t mdh_dense_passes(const mdh_ctx* ctx);   /* batched passes queued in the dense format so far */
int mdh_dense_now(const mdh_ctx* ctx);          /* 1: the next batched pass will be queued in the dense format */
double mdh_last_density(const mdh_ctx* ctx);    /* last collected pass: share of frames with a dense binary (sparse pass) or of words with set bits (dense pass) */

/* The binary image `dst` (H*W u8 of {0,255}) of frame `index` of the most recent pass (0 .. frames_in_pass-1;
 * -1 = the last frame).  For visu() and parity tests.  stage: 0 = dst (after the dynamic mask), 1 = the image
 * before the dynamic mask (MORPH_CLOSE output). */
int mdh_get_binary(mdh_ctx* ctx, int index, int stage, uint8_t* out);

/* lineset_nms (utils.py:780-839) on the host.  lines: n x 4 int32.  Returns the number of kept segments. */
int mdh_lineset_nms(const int32_t* lines, int n, int32_t* out_lines, double* nonline);

/* Device-memory helpers so a harness can keep synthetic frames resident in HBM. */
void* mdh_device_alloc(mdh_ctx* ctx, uint64_t bytes);
int mdh_device_free(mdh_ctx* ctx, void* ptr);
int mdh_device_upload(mdh_ctx* ctx, void* dst_device, const void* src_host, uint64_t bytes);
/* device-to-device copy between buffers of this context's GPU (the single-process stand-in of the bit-plane exchange) */
int mdh_device_copy(mdh_ctx* ctx, void* dst_device, const void* src_device, uint64_t bytes);
/* Pinned host memory for frames handed over with on_device = 0: the H2D copies of a pass then run asynchronously at PCIe
 * speed (from pageable memory they are staged by the runtime). */
void* mdh_host_alloc(mdh_ctx* ctx, uint64_t bytes);
int mdh_host_free(mdh_ctx* ctx, void* ptr);   /* a pool may be freed through any context of the process, never by hipHostFree directly */
/* The per-frame protocol reads a frame IN PLACE (no copy in front of the window kernel) when the caller's buffer is memory a
 * kernel can reach AND the library knows about: a pool of mdh_host_alloc, or a range announced with mdh_host_register -- ordinary
 * memory (registered here) or memory the caller pinned itself (hipHostMalloc / hipHostRegister of its own: accepted as it is since
 * round 6, and left to the caller to release).  Up to round 5 the library also asked the runtime about any other pointer
 * (hipPointerGetAttributes); the runtime gives the same answer for a range that was registered once and freed without being
 * unregistered, and a read through such a stale mapping is a GPU memory fault that aborts the process.  The reference's loaders hand over plain NumPy arrays (MetDetPy.py:192; videoloader.py:569-593), and
 * ProcessVideoLoader recycles a fixed ring of them (a shared RawArray, videoloader.py:696-701, 746-772):
 *   mdh_host_register(ctx, ring_base, ring_bytes)   once, by a loader that knows its ring;  mdh_host_unregister before freeing it
 *   mdh_set_auto_register(ctx, 1)  (OFF by default since ABI 4; MDH_AUTO_REGISTER=1 in the environment switches it on): the
 *       third time mdh_update is handed the same pageable address the library registers that frame buffer itself (at most 64
 *       buffers per context, unregistered by mdh_destroy).  Opt-in because it pins memory the CALLER owns: a caller that switches
 *       it on promises to call mdh_host_unregister before it frees (or unmaps) such a buffer while the context lives -- a loader
 *       that hands over a fresh array per frame gets the same malloc address back again and again and cannot keep that promise,
 *       so it must not be enrolled behind its back (a stale registration reads freed pages).
 * Frames in memory no kernel can reach are copied (H2D from pageable memory); results are identical. */
int mdh_host_register(mdh_ctx* ctx, void* ptr, uint64_t bytes);
int mdh_host_unregister(mdh_ctx* ctx, void* ptr);
int mdh_set_auto_register(mdh_ctx* ctx, int on);
/* mdh_update so far: frames the window kernel read in place / frames copied first / buffers registered by the library */
int mdh_update_stats(const mdh_ctx* ctx, int64_t* direct_frames, int64_t* copied_frames, int* auto_registered);

/* Per-kernel timing with HIP events on the context's own stream (off by default).
 * names: MDH_NUM_KERNELS strings; ms / launches: accumulated since the last reset. */
#define MDH_NUM_KERNELS 8
int mdh_profile_enable(mdh_ctx* ctx, int on);   /* on = N > 0: every N-th pass is timed (1 = all); 0 = off */
int mdh_profile_read(mdh_ctx* ctx, double* ms, int64_t* launches, int reset);
const char* mdh_kernel_name(int k);

/* Loader-side preprocessing in front of the detector:
 *   Transform.exec_transform = cv2.resize(INTER_LINEAR) -> cv2.cvtColor(BGR2GRAY) -> img * mask   imgproc.py:82-101, 129-139
 *   MergeFunction.max over exp_frame consecutive processed frames                                utils.py:202-204
 * bgr: exp_frame x src_h x src_w x 3 u8 (host or device); out: dst_h x dst_w u8 (host or device).
 * mask01 (dst_h x dst_w of {0,1}, as fileio.load_mask returns) may be NULL. */
typedef struct mdh_pre mdh_pre;
mdh_pre* mdh_pre_create(int device, int src_h, int src_w, int dst_h, int dst_w, const uint8_t* mask01);
int mdh_pre_run(mdh_pre* pre, const uint8_t* bgr, int exp_frame, int src_on_device, uint8_t* out, int out_on_device);
/* n_out output frames in one launch: bgr holds n_out * exp_frame source frames, out receives n_out * dst_h * dst_w bytes --
 * with out_on_device = 1 a buffer that mdh_submit_batch / mdh_process_batch consume in place (on_device = 1), so decoded
 * frames reach the detector without leaving HBM. */
int mdh_pre_run_batch(mdh_pre* pre, const uint8_t* bgr, int n_out, int exp_frame, int src_on_device, uint8_t* out, int out_on_device);
void mdh_pre_destroy(mdh_pre* pre);
/* bytes a launch must move per output frame: the source rows the resize touches (exp_frame frames) + mask + output */
uint64_t mdh_pre_traffic_model(const mdh_pre* pre, int exp_frame);

/* stacker.max_stacker / MaxImgContainer (stacker.py:43-49, 197-213): elementwise max over K frames of frame_bytes
 * bytes each (any channel layout). */
int mdh_max_stack(int device, const uint8_t* frames, int K, uint64_t frame_bytes, int on_device, uint8_t* out,
                  int out_on_device);

/* np.sum(frame) for K frames of frame_bytes bytes each (a multiple of 16, 16-byte aligned): the per-frame brightness series
 * exposure estimation works on (_rf_est_kernel, videoloader.py:796-819).  sums: K uint64, host memory. */
int mdh_frame_sums(int device, const uint8_t* frames, int K, uint64_t frame_bytes, int on_device, uint64_t* sums);

/* ---- Row-band tiling of ONE stream across GPUs (BASELINE config 5; SURVEY 8e "spatial row-bands") ----------------
 *
 * The frame is cut into nbands bands of band_rows = ceil(H / nbands) rows, one per GPU.  Band b owns rows
 * [b*band_rows, min(H, (b+1)*band_rows)) and is created (role MDH_ROLE_BAND) on the sub-image made of those rows plus
 * MDH_BAND_HALO rows of each existing neighbour: every stage of M3Detector.detect before the Hough transform is a 3x3
 * stencil chain (median, dilate, erode, dynamic-mask erode: Detector.py:329-340, 234-242), so a sub-image processed as
 * a standalone frame is wrong only within 4 rows of an artificial border, i.e. only inside the halo.  Its cfg.roi is
 * the part of SNR_SW.std_roi (Detector.py:93-122) inside the OWNED rows, in sub-image coordinates (empty when the band
 * holds none: roi[2] == roi[0]); cfg.mask_area is the whole frame's.  Per pass of K frames:
 *
 *   1. mdh_band_noise      this band's share of the noise sums of the due frames (Detector.py:81-91)
 *        -> all-reduce(sum) of sums[0 .. 2*n_jobs) over the bands
 *   2. mdh_band_pixels     thresholds from the global sums, pixel stages, owned rows of the dst bit planes
 *        -> all-to-all: frame i's rows go to the rank that runs its Hough stage (rank i % nsplit)
 *   3. mdh_hough_planes    (MDH_ROLE_HOUGH context of the whole frame, any rank) lit count, adaptive gap,
 *                          HoughLinesP, lineset_nms for the frames whose pieces it received
 *
 * All three calls block until their results are ready. */

/* Step 1.  frames: K * H * W u8 (this context's sub-image), host or device.  sums: 2 * K int64 (S1, S2 per due frame,
 * in frame order); *n_jobs = number of due frames (identical on every band, it depends on the frame counter only). */
int mdh_band_noise(mdh_ctx* ctx, const uint8_t* frames, int K, int on_device, int64_t* sums, int* n_jobs);

/* Step 2.  sums_total: the all-reduced sums; roi_px_total: pixels of the whole std_roi.  Rows [own_row0, own_row0 +
 * own_rows) of frame i's dst bit plane (WW = ceil(W/64) u64 words per row, bit k of word w = pixel 64w+k) are written
 * to dst_rows + (slot(i) * rows_stride) * WW, the rows own_rows .. rows_stride-1 of each slot zero-filled, where
 * slot(i) = (i % nsplit) * ceil(K / nsplit) + i / nsplit -- the send layout of an equal-split all-to-all in which rank r
 * receives frames r, r + nsplit, ...; nsplit = 1 gives frame order.  dst_rows: nsplit * ceil(K/nsplit) * rows_stride *
 * WW u64, host or device.  info (K entries, may be NULL): bi_threshold, bi_threshold_float, snr, sigma filled. */
int mdh_band_pixels(mdh_ctx* ctx, const int64_t* sums_total, double roi_px_total, int own_row0, int own_rows,
                    int rows_stride, int nsplit, uint64_t* dst_rows, int out_on_device, mdh_frame_info* info);

/* Step 3.  pieces: nbands * K * band_rows * WW u64; piece b of frame i (rows [b*band_rows, (b+1)*band_rows) of the
 * frame, rows beyond H ignored) starts at ((b * K + i) * band_rows) * WW -- the receive layout of the all-to-all above.
 * Outputs as mdh_process_batch (bi_threshold / snr fields of info are left 0: they come from step 2). */
int mdh_hough_planes(mdh_ctx* ctx, const uint64_t* pieces, int K, int on_device, int nbands, int band_rows,
                     mdh_frame_info* info, int32_t* raw_lines, int32_t* nms_lines, double* nonline);
/* The same, asynchronous (the Hough stage of pass p then runs under the band stages of pass p + 1): queue the pass and return;
 * mdh_collect_batch hands out the results of the oldest queued pass.  Pieces in device memory must stay untouched until the
 * pass after the next one is submitted or this one is collected (two exchange buffers used alternately are enough). */
int mdh_hough_planes_submit(mdh_ctx* ctx, const uint64_t* pieces, int K, int on_device, int nbands, int band_rows);

/* The same three steps WITHOUT host round trips (round 3).  Every call below only queues work; the sums never leave the device:
 *
 *   mdh_band_submit(band, frames, K, on_device, &n_jobs)          stage + this band's share of the noise sums (device memory)
 *   all-reduce IN PLACE of mdh_band_sums_device(band) [2 * n_jobs int64] on stream mdh_stream_handle(band, MDH_STREAM_NOISE):
 *       ncclAllReduce(p, p, 2 * n_jobs, ncclInt64, ncclSum, comm, stream)   (RCCL, one band per GPU; metdetpy_amd/rccl.py)
 *       mdh_bands_allreduce_local(bands, n)                                 (all bands in one process on one GPU)
 *   mdh_band_pixels_submit(band, NULL, roi_px_total, ..., send_buffer)     thresholds, pixel stages, owned rows -> send layout
 *   all-to-all of the send buffer on stream mdh_stream_handle(band, MDH_STREAM_POST_LAST):
 *       ncclGroupStart(); ncclSend / ncclRecv per peer; ncclGroupEnd()     (RCCL)
 *       mdh_chain(band_b, MDH_STREAM_POST_LAST, hough_r, MDH_STREAM_POST_NEXT) + mdh_device_copy_on(hough_r, MDH_STREAM_POST_NEXT, ...)
 *   mdh_chain(band, MDH_STREAM_POST_LAST, hough, MDH_STREAM_POST_NEXT); mdh_hough_planes_submit(hough, recv_buffer, ...)
 *   ... next pass ...
 *   mdh_band_collect(band, info)  (thresholds of the oldest band pass)   +   mdh_collect_batch(hough, ...)  (its Hough results)
 *
 * Buffers handed to a queued pass (send / receive) must stay untouched until that pass has been collected. */
#define MDH_STREAM_NOISE 0       /* staging + noise / threshold chain */
#define MDH_STREAM_POST_LAST 1   /* post-window stages of the pass queued last (a band context's exported rows come out of it) */
#define MDH_STREAM_POST_NEXT 2   /* post-window stages of the pass queued next (a Hough context assembles its planes on it) */
#define MDH_MAX_LOCAL_BANDS 16
void* mdh_stream_handle(mdh_ctx* ctx, int which);   /* hipStream_t, for collectives enqueued by the host (RCCL) */
int mdh_band_submit(mdh_ctx* ctx, const uint8_t* frames, int K, int on_device, int* n_jobs);
void* mdh_band_sums_device(mdh_ctx* ctx);           /* device int64 [2 * n_jobs] of the pass staged by mdh_band_submit */
/* sums_total_device: NULL (the sums were all-reduced in place) or a device pointer to 2 * n_jobs int64.  dst_rows_device: as
 * mdh_band_pixels' dst_rows with out_on_device = 1. */
int mdh_band_pixels_submit(mdh_ctx* ctx, const int64_t* sums_total_device, double roi_px_total, int own_row0, int own_rows,
                           int rows_stride, int nsplit, uint64_t* dst_rows_device, int flags);
/* The thresholds of the staged pass on their own (the sums as for mdh_band_pixels_submit, which then skips this part).  For a
 * process that drives several bands on one device: queued for ALL bands before the first band's pixel stages, the scalar chains run
 * while the device is quiet instead of each waiting for a CU between two bands' window kernels. */
int mdh_band_thresholds_submit(mdh_ctx* ctx, const int64_t* sums_total_device, double roi_px_total);
#define MDH_EXPORT_NOFILL 1     /* flags: do not zero-fill the slots first (the destination is a plane several bands write into) */
/* All bands in ONE process: the dst planes ([K][H][WW] u64) the next pass queued on a MDH_ROLE_HOUGH context will read.  Every
 * band exports its rows straight into them (dst_rows_device = planes + own_first_row * WW, rows_stride = H, nsplit = 1,
 * MDH_EXPORT_NOFILL; together the bands cover every row) and mdh_hough_planes_submit(ctx, planes, K, 1, 1, H) assembles in
 * place: the exchange costs nothing. */
void* mdh_hough_input_planes(mdh_ctx* ctx);
/* The same in one call, counters included: band `ctx` (thresholds as for mdh_band_pixels_submit) exports its owned rows into the
 * planes of the pass `hough` queues next, rows full_row0 ..., and adds their row / frame lit counts and bounding box to that pass's
 * counters (the first band of a pass zeroes them), so that mdh_hough_planes_submit(hough, mdh_hough_input_planes(hough), K, 1, 1, H)
 * has nothing left to assemble or count.  Every band of the pass must use this entry point; mdh_chain the band's post stream
 * into the Hough context before submitting, as with mdh_band_pixels_submit. */
int mdh_band_pixels_submit_to(mdh_ctx* ctx, const int64_t* sums_total_device, double roi_px_total, int own_row0, int own_rows,
                              mdh_ctx* hough, int full_row0);
int mdh_band_collect(mdh_ctx* ctx, mdh_frame_info* info);   /* waits for the oldest queued band pass; returns its frame count */
/* everything queued on producer's stream so far precedes whatever is queued on consumer's stream from now on (same process) */
int mdh_chain(mdh_ctx* producer, int producer_stream, mdh_ctx* consumer, int consumer_stream);
int mdh_device_copy_on(mdh_ctx* ctx, int which_stream, void* dst_device, const void* src_device, uint64_t bytes);  /* queued */
int mdh_bands_allreduce_local(mdh_ctx** bands, int n);      /* n band contexts of one process and device, passes staged */
/* Several contexts of one process on one device queue their stages on the SAME streams (ctx gives up its own main / noise / post
 * streams and uses donor's): the runtime multiplexes streams onto a few hardware queues, and unrelated streams sharing a queue
 * serialise.  Right after mdh_create, before ctx's first pass; donor must outlive ctx. */
int mdh_share_streams(mdh_ctx* ctx, mdh_ctx* donor);
/* How long mdh_collect_batch / mdh_band_collect / mdh_detect wait for their pass (0, the default: for ever).  A queued pass that
 * contains a collective never completes when a peer rank has died; with a timeout the wait returns MDH_E_TIMEOUT and poisons the
 * context -- the host then aborts its communicator (ncclCommAbort releases the stuck kernels) and destroys the context. */
int mdh_set_wait_timeout_ms(mdh_ctx* ctx, double ms);

/* ---- One stream cut into time windows across GPUs (SURVEY 8e "time windows of one stream") --------------------------
 *
 * Rank r processes frames [s_r, e_r) of the stream, starting 2(n-1) frames early (n-1 to fill the max/mean window of the
 * first binaries the dynamic mask needs, n-1 more for those binaries themselves); the results of the warm-up frames are
 * dropped.  What couples the chunks is the threshold: T(t) follows an EMA of the noise sigma with a long memory
 * (Detector.py:61-62, utils.py:334-368).  The sigma of a due frame depends only on the n frames before it, so every
 * rank computes the noise sums of the due frames of its own chunk (mdh_band_noise + mdh_band_skip over the chunk),
 * the sums are all-gathered (16 bytes per due frame), every rank replays the scalar chain on the host up to its start,
 * positions its context there (mdh_seek) and then runs the chunk through mdh_band_noise / mdh_band_pixels (one band =
 * the whole frame, sums from the gathered table) and mdh_hough_planes.  metdetpy_amd/timeshard.py drives it. */

/* Drop the pass staged by mdh_band_noise without running its pixel stages (the frames stay in the window). */
int mdh_band_skip(mdh_ctx* ctx);

/* Reposition a context in its stream: the next frame is number t0 + 1 (SlidingWindow.timer == t0), the window and the
 * ring of binaries are empty, the noise EMA has been updated ema_updates times and stands at snr (EMA.cur_value), with
 * sigma / thr_float / thr the values in force after frame t0.  No pass may be in flight.  The 2(n-1) frames that follow are
 * warm-up (their results are meaningless: the window, then the ring of binaries, is still filling); for t0 > 0 the batched M3
 * passes leave the binaries of the first n-1 of them empty instead of thresholding windows that hold zero frames. */
int mdh_seek(mdh_ctx* ctx, int64_t t0, int64_t ema_updates, double snr, double sigma, double thr_float, int thr);

/* Time windows on a MDH_ROLE_FULL context (the pipelined mdh_submit_batch / mdh_collect_batch path, thresholds computed by the
 * context itself): after mdh_seek the window is empty, so a sigma that falls due inside the 2(n-1) warm-up frames would be
 * taken over an incomplete window -- and would then sit in the EMA for the rest of the chunk.  mdh_inject_sums supplies the
 * sums (S1, S2: Detector.py:81-91 over the sub-ROI, as mdh_band_noise returns them) of due frame number t (1-based, not yet
 * submitted) from the all-gathered table; the chain uses them instead of its own when frame t comes by.  Cleared by mdh_seek.
 * mdh_mute_hough_until: frames numbered <= t skip the Hough stage (n_raw = 0): a window that starts empty turns its first
 * frames into dense binaries whose replay would dominate the chunk, and their results are dropped anyway. */
int mdh_inject_sums(mdh_ctx* ctx, int64_t t, int64_t s1, int64_t s2);
int mdh_mute_hough_until(mdh_ctx* ctx, int64_t t);

/* ---- Native MeteorCollector bookkeeping (collector.py:372-631) ---------------------------------------------
 * Host code, no GPU needed.  Consumes (frame, lines, cates) exactly as MeteorCollector.update does and queues the series it
 * finishes as mdh_target records (unrounded); metdetpy_amd/collector.py turns them into MDRF records. */
typedef struct mdh_collector_cfg {
    uint32_t struct_size;                /* sizeof(mdh_collector_cfg): checked first by mdh_collector_create */
    uint32_t reserved;                   /* 0 */
    double min_len, max_interval;        /* MeteorCfg.min_len, .max_interval (seconds) */
    double time_range[2], speed_range[2], drct_range[2];
    double det_thre;
    double thre2;                        /* MeteorCfg.thre2 (multiplied by exp_frame inside, collector.py:403) */
    double fps;                          /* RuntimeParams.fps */
    int32_t exp_frame;
    int32_t runtime_w, runtime_h;        /* RuntimeParams.runtime_size */
    int32_t num_cls;
} mdh_collector_cfg;

typedef struct mdh_target {              /* MeteorSeries.get_met_attr + score, before rounding (collector.py:283-306, 617-627) */
    int32_t flag;                        /* 0: dropped series (DROP_FLAG), 1: member of an exported waiting list (ACTIVE_FLAG) */
    int32_t group;                       /* records of one export() call share a group number, in export order */
    int64_t start_frame, end_frame, last_activate_frame, duration;
    double speed, dist, fix_dist, fix_speed, fix_motion_duration, fix_duration, drct_loss, score;
    int32_t num_pts, category;           /* category: class id (arg-max of the accumulated class probabilities) */
    int32_t pt1[2], pt2[2];
    int32_t center_off, n_center;        /* centre points: centers[2*(center_off + j)], [.. + 1], j < n_center (float64 means) */
    int32_t score_is_int;                /* the reference's score is a Python int here (all four factors on flat parts of their curves) */
    int32_t reserved;
} mdh_target;

typedef struct mdh_collector mdh_collector;
mdh_collector* mdh_collector_create(const mdh_collector_cfg* cfg);
void mdh_collector_destroy(mdh_collector* c);
/* MeteorCollector.update(cur_frame, lines, cates): lines n x 4 int32, cates n x num_cls float64 (n may be 0). */
int mdh_collector_update(mdh_collector* c, int64_t cur_frame, const int32_t* lines, const double* cates, int n);
/* K frames of one detector pass in one call, with the main loop's gating (MetDetPy.py:200-202): frame i of the pass is
 * iteration it0 + i and video frame frame0 + i * exp_frame; it is fed when it has lines or its iteration is a multiple of
 * eq_int_fps.  info / nms_lines / nonline: as mdh_collect_batch returns them (cls_pred is rebuilt as column 0 = 1 - nonline,
 * last column = nonline, Detector.py:382-392). */
int mdh_collector_update_batch(mdh_collector* c, int K, int64_t frame0, int64_t it0, int eq_int_fps, const mdh_frame_info* info,
                               const int32_t* nms_lines, const double* nonline);
/* Queued records: returns their number (negative on error) and copies up to cap of them / cap_centers centre points; the
 * queue is emptied only when everything fitted (call with cap = 0 to size the buffers: *n_centers is always set). */
int mdh_collector_events(mdh_collector* c, mdh_target* out, int cap, double* centers, int cap_centers, int* n_centers);
int mdh_collector_active(const mdh_collector* c);   /* series currently open (without the reference's sentinel) */

/* Debug: with MDH_HOUGH_DEBUG=1 in the environment at mdh_create, the Hough kernels record per frame of the last pass 8
 * stamps (lit pixels, tier / list build, set-up, PPHT total, ...) and 8 phase totals of the four-wave kernel in shader
 * cycles (draws + conflict scan, list exchange, mask checks, votes, keys + mailbox, trigger handling, line handling, groups).
 * out: K*8 stamps followed by K*8 phase totals. */
int mdh_debug_hough(mdh_ctx* ctx, long long* out, int K);

/* Frames consumed so far (SlidingWindow.timer). */
int64_t mdh_frames_seen(const mdh_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* METDET_HIP_H */

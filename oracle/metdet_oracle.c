/* CPU oracle (C port) for the MetDetPy line-detector hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the same arithmetic as oracle/m3_oracle.py, fast enough to check the
 * HIP path at full frame sizes and to serve as the `cpu_baseline` ("port") leg of bench.py.
 * Nothing in the product package links or loads this file.
 *
 * Structure-faithful to the reference: an n-deep ring re-reduced with a max over the whole ring on
 * every frame (utils.py:269-307), a second n-deep ring of binary images for the dynamic mask
 * (Detector.py:234-242), the noise sigma over all ring samples of the sub-ROI (Detector.py:81-91).
 * OpenCV-side ops follow SURVEY.md Appendix A; they are PARITY UNPINNED (cv2 is neither vendored in
 * the reference nor installable here) -- see oracle/m3_oracle.py's header.
 *
 * Build: make -C oracle    (gcc -O3 -march=x86-64-v3 -fopenmp -ffp-contract=off)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NUM_ANGLE 180

/* ------------------------------------------------------------------ 3x3 integer image ops */
static inline uint8_t px_clamped(const uint8_t* img, int H, int W, int y, int x) {
    if (y < 0) y = 0;
    if (y >= H) y = H - 1;
    if (x < 0) x = 0;
    if (x >= W) x = W - 1;
    return img[(size_t)y * W + x];
}

/* cv2.medianBlur(src, 3): exact median of 9, replicated border (Detector.py:329). */
void ocv_median3(const uint8_t* src, uint8_t* dst, int H, int W) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint8_t v[9];
            int k = 0;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) v[k++] = px_clamped(src, H, W, y + dy, x + dx);
            for (int i = 1; i < 9; ++i) { /* insertion sort */
                uint8_t t = v[i];
                int j = i - 1;
                while (j >= 0 && v[j] > t) { v[j + 1] = v[j]; --j; }
                v[j + 1] = t;
            }
            dst[(size_t)y * W + x] = v[4];
        }
}

/* 3x3 max / min over the in-image part of the window (cv2.dilate / cv2.erode, default border). */
static void minmax3(const uint8_t* src, uint8_t* dst, int H, int W, int want_max) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            int best = want_max ? 0 : 255;
            for (int dy = -1; dy <= 1; ++dy) {
                int yy = y + dy;
                if (yy < 0 || yy >= H) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    int xx = x + dx;
                    if (xx < 0 || xx >= W) continue;
                    int v = src[(size_t)yy * W + xx];
                    if (want_max ? v > best : v < best) best = v;
                }
            }
            dst[(size_t)y * W + x] = (uint8_t)best;
        }
}
void ocv_dilate3(const uint8_t* src, uint8_t* dst, int H, int W) { minmax3(src, dst, H, W, 1); }
void ocv_erode3(const uint8_t* src, uint8_t* dst, int H, int W) { minmax3(src, dst, H, W, 0); }

/* ------------------------------------------------------------------ cv::RNG + HoughLinesP */
typedef struct { uint64_t state; } cv_rng;
static inline uint32_t rng_next(cv_rng* r) {
    r->state = (uint64_t)(uint32_t)r->state * 4164903690u + (r->state >> 32);
    return (uint32_t)r->state;
}

/* cv2.HoughLinesP(img, 1, (float)(pi/180), threshold, minLineLength, maxLineGap)
 * (Detector.py:282-287, 347-352; OpenCV hough.cpp HoughLinesProbabilistic).
 * lines: cap x 4 int32 [x1,y1,x2,y2]; returns the TOTAL number of segments found (may exceed cap;
 * only the first cap are stored).  Scratch is allocated per call. */
int ocv_hough_lines_p(const uint8_t* img, int H, int W, int threshold, double min_len, double max_gap,
                      int32_t* lines, int cap) {
    const int line_length = (int)lrint(min_len);
    const int line_gap = (int)lrint(max_gap);
    const float theta = (float)(3.14159265358979323846 / 180.0);
    const int numrho = (W + H) * 2 + 1;
    const int half = (numrho - 1) / 2;
    const int shift = 16;
    float tcos[NUM_ANGLE], tsin[NUM_ANGLE];
    for (int n = 0; n < NUM_ANGLE; ++n) {
        tcos[n] = (float)(cos((double)n * theta) * 1.0f);
        tsin[n] = (float)(sin((double)n * theta) * 1.0f);
    }
    int32_t* accum = (int32_t*)calloc((size_t)NUM_ANGLE * numrho, sizeof(int32_t));
    uint8_t* mask = (uint8_t*)malloc((size_t)H * W);
    int count = 0;
    for (size_t p = 0; p < (size_t)H * W; ++p) {
        mask[p] = img[p] != 0;
        count += mask[p];
    }
    int32_t* nz = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(count ? count : 1));
    {
        int k = 0;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                if (mask[(size_t)y * W + x]) { nz[2 * k] = x; nz[2 * k + 1] = y; ++k; }
    }
    cv_rng rng = {0xFFFFFFFFFFFFFFFFull};
    int found = 0;
    for (; count > 0; --count) {
        int idx = (int)(rng_next(&rng) % (uint32_t)count);
        int j = nz[2 * idx], i = nz[2 * idx + 1];
        nz[2 * idx] = nz[2 * (count - 1)];
        nz[2 * idx + 1] = nz[2 * (count - 1) + 1];
        if (!mask[(size_t)i * W + j]) continue;
        int max_val = threshold - 1, max_n = 0;
        for (int n = 0; n < NUM_ANGLE; ++n) {
            int r = (int)lrintf((float)j * tcos[n] + (float)i * tsin[n]) + half;
            int val = ++accum[(size_t)n * numrho + r];
            if (max_val < val) { max_val = val; max_n = n; }
        }
        if (max_val < threshold) continue;
        float a = -tsin[max_n], b = tcos[max_n];
        int x0 = j, y0 = i, dx0, dy0, xflag;
        if (fabsf(a) > fabsf(b)) {
            xflag = 1;
            dx0 = a > 0 ? 1 : -1;
            dy0 = (int)lrintf(b * (float)(1 << shift) / fabsf(a));
            y0 = (y0 << shift) + (1 << (shift - 1));
        } else {
            xflag = 0;
            dy0 = b > 0 ? 1 : -1;
            dx0 = (int)lrintf(a * (float)(1 << shift) / fabsf(b));
            x0 = (x0 << shift) + (1 << (shift - 1));
        }
        int end_x[2] = {0, 0}, end_y[2] = {0, 0};
        for (int k = 0; k < 2; ++k) {
            int gap = 0, x = x0, y = y0, dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            for (;; x += dx, y += dy) {
                int j1 = xflag ? x : x >> shift, i1 = xflag ? y >> shift : y;
                if (j1 < 0 || j1 >= W || i1 < 0 || i1 >= H) break;
                if (mask[(size_t)i1 * W + j1]) {
                    gap = 0;
                    end_x[k] = j1;
                    end_y[k] = i1;
                } else if (++gap > line_gap)
                    break;
            }
        }
        int good = abs(end_x[1] - end_x[0]) >= line_length || abs(end_y[1] - end_y[0]) >= line_length;
        for (int k = 0; k < 2; ++k) {
            int x = x0, y = y0, dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            for (;; x += dx, y += dy) {
                int j1 = xflag ? x : x >> shift, i1 = xflag ? y >> shift : y;
                uint8_t* m = &mask[(size_t)i1 * W + j1];
                if (*m) {
                    if (good)
                        for (int n = 0; n < NUM_ANGLE; ++n) {
                            int r = (int)lrintf((float)j1 * tcos[n] + (float)i1 * tsin[n]) + half;
                            accum[(size_t)n * numrho + r]--;
                        }
                    *m = 0;
                }
                if (i1 == end_y[k] && j1 == end_x[k]) break;
            }
        }
        if (good) {
            if (found < cap) {
                lines[4 * found + 0] = end_x[0];
                lines[4 * found + 1] = end_y[0];
                lines[4 * found + 2] = end_x[1];
                lines[4 * found + 3] = end_y[1];
            }
            ++found;
        }
    }
    free(nz);
    free(mask);
    free(accum);
    return found;
}

/* ------------------------------------------------------------------ hough_mode = accumulator
 * C port of oracle/hough_accumulator.py (the specification of MDH_HOUGH_ACCUMULATOR; NOT cv2.HoughLinesP): all lit pixels vote,
 * then repeatedly the largest cell (smaller angle, then smaller rho among equals) while it holds >= threshold votes; seed = the
 * middle one (row-major) of the lit pixels voting for it; OpenCV's walk / ends / good test; every lit pixel on the walk is
 * cleared and un-voted whether or not the segment is good.  More than 8192 lit pixels: the exact progressive transform.
 * Per-angle maxima are cached and rescanned only when an un-vote hit the cached cell (an optimisation with the same result:
 * other cells of the row only shrink). */
#define OACC_MAX_LIST 8192
int oacc_hough_lines(const uint8_t* img, int H, int W, int threshold, double min_len, double max_gap, int32_t* lines, int cap) {
    const int line_length = (int)lrint(min_len);
    const int line_gap = (int)lrint(max_gap);
    const float theta = (float)(3.14159265358979323846 / 180.0);
    const int numrho = (W + H) * 2 + 1;
    const int half = (numrho - 1) / 2;
    const int shift = 16;
    int N = 0;
    for (size_t p = 0; p < (size_t)H * W; ++p) N += img[p] != 0;
    if (N < threshold) return 0;
    if (N > OACC_MAX_LIST) return ocv_hough_lines_p(img, H, W, threshold, min_len, max_gap, lines, cap);
    float tcos[NUM_ANGLE], tsin[NUM_ANGLE];
    for (int n = 0; n < NUM_ANGLE; ++n) {
        tcos[n] = (float)(cos((double)n * theta) * 1.0f);
        tsin[n] = (float)(sin((double)n * theta) * 1.0f);
    }
    int32_t* accum = (int32_t*)calloc((size_t)NUM_ANGLE * numrho, sizeof(int32_t));
    uint8_t* mask = (uint8_t*)malloc((size_t)H * W);
    int32_t* nz = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)N);
    {
        int k = 0;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                mask[(size_t)y * W + x] = img[(size_t)y * W + x] != 0;
                if (mask[(size_t)y * W + x]) { nz[2 * k] = x; nz[2 * k + 1] = y; ++k; }
            }
    }
    int lo[NUM_ANGLE], hi[NUM_ANGLE], best[NUM_ANGLE], best_r[NUM_ANGLE], dirty[NUM_ANGLE];
    for (int n = 0; n < NUM_ANGLE; ++n) { lo[n] = numrho; hi[n] = -1; dirty[n] = 1; best[n] = -1; best_r[n] = 0; }
    for (int k = 0; k < N; ++k)
        for (int n = 0; n < NUM_ANGLE; ++n) {
            int r = (int)lrintf((float)nz[2 * k] * tcos[n] + (float)nz[2 * k + 1] * tsin[n]) + half;
            accum[(size_t)n * numrho + r]++;
            if (r < lo[n]) lo[n] = r;
            if (r > hi[n]) hi[n] = r;
        }
    int found = 0;
    for (int iter = 0; iter <= N; ++iter) {
        int top = -1, bn = 0;
        for (int n = 0; n < NUM_ANGLE; ++n) {
            if (dirty[n]) {
                best[n] = -1;
                best_r[n] = lo[n];
                for (int r = lo[n]; r <= hi[n]; ++r)
                    if (accum[(size_t)n * numrho + r] > best[n]) { best[n] = accum[(size_t)n * numrho + r]; best_r[n] = r; }
                dirty[n] = 0;
            }
            if (best[n] > top) { top = best[n]; bn = n; }
        }
        if (top < threshold) break;
        const int rho = best_r[bn];
        int voters = 0;
        for (int k = 0; k < N; ++k) {
            int x = nz[2 * k], y = nz[2 * k + 1];
            if (mask[(size_t)y * W + x] && (int)lrintf((float)x * tcos[bn] + (float)y * tsin[bn]) + half == rho) ++voters;
        }
        int want = voters / 2, j = 0, i = 0;
        for (int k = 0; k < N; ++k) {
            int x = nz[2 * k], y = nz[2 * k + 1];
            if (mask[(size_t)y * W + x] && (int)lrintf((float)x * tcos[bn] + (float)y * tsin[bn]) + half == rho && want-- == 0) { j = x; i = y; break; }
        }
        float a = -tsin[bn], b = tcos[bn];
        int x0 = j, y0 = i, dx0, dy0, xflag;
        if (fabsf(a) > fabsf(b)) {
            xflag = 1;
            dx0 = a > 0 ? 1 : -1;
            dy0 = (int)lrintf(b * (float)(1 << shift) / fabsf(a));
            y0 = (y0 << shift) + (1 << (shift - 1));
        } else {
            xflag = 0;
            dy0 = b > 0 ? 1 : -1;
            dx0 = (int)lrintf(a * (float)(1 << shift) / fabsf(b));
            x0 = (x0 << shift) + (1 << (shift - 1));
        }
        int end_x[2] = {0, 0}, end_y[2] = {0, 0};
        for (int k = 0; k < 2; ++k) {
            int gap = 0, x = x0, y = y0, dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            for (;; x += dx, y += dy) {
                int j1 = xflag ? x : x >> shift, i1 = xflag ? y >> shift : y;
                if (j1 < 0 || j1 >= W || i1 < 0 || i1 >= H) break;
                if (mask[(size_t)i1 * W + j1]) {
                    gap = 0;
                    end_x[k] = j1;
                    end_y[k] = i1;
                } else if (++gap > line_gap)
                    break;
            }
        }
        int good = abs(end_x[1] - end_x[0]) >= line_length || abs(end_y[1] - end_y[0]) >= line_length;
        for (int k = 0; k < 2; ++k) {
            int x = x0, y = y0, dx = k ? -dx0 : dx0, dy = k ? -dy0 : dy0;
            for (;; x += dx, y += dy) {
                int j1 = xflag ? x : x >> shift, i1 = xflag ? y >> shift : y;
                uint8_t* m = &mask[(size_t)i1 * W + j1];
                if (*m) {
                    for (int n = 0; n < NUM_ANGLE; ++n) {
                        int r = (int)lrintf((float)j1 * tcos[n] + (float)i1 * tsin[n]) + half;
                        accum[(size_t)n * numrho + r]--;
                        if (r == best_r[n]) dirty[n] = 1;
                    }
                    *m = 0;
                }
                if (i1 == end_y[k] && j1 == end_x[k]) break;
            }
        }
        if (good) {
            if (found < cap) {
                lines[4 * found + 0] = end_x[0];
                lines[4 * found + 1] = end_y[0];
                lines[4 * found + 2] = end_x[1];
                lines[4 * found + 3] = end_y[1];
            }
            ++found;
        }
    }
    free(nz);
    free(mask);
    free(accum);
    return found;
}

/* ------------------------------------------------------------------ lineset_nms (utils.py:780-839) */
/* Stable ascending order on squared length, reversed (later index first among ties). */
int o_lineset_nms(const int32_t* lines, int n, int32_t* out_lines, double* nonline) {
    if (n <= 0) return 0;
    int64_t* len2 = (int64_t*)malloc(sizeof(int64_t) * n);
    int* order = (int*)malloc(sizeof(int) * n);
    uint8_t* taken = (uint8_t*)calloc(n, 1);
    for (int i = 0; i < n; ++i) {
        int64_t dx = lines[4 * i + 2] - lines[4 * i + 0], dy = lines[4 * i + 3] - lines[4 * i + 1];
        len2[i] = dx * dx + dy * dy;
        order[i] = i;
    }
    for (int i = 1; i < n; ++i) { /* stable insertion sort, ascending */
        int t = order[i], j = i - 1;
        while (j >= 0 && len2[order[j]] > len2[t]) { order[j + 1] = order[j]; --j; }
        order[j + 1] = t;
    }
    for (int i = 0; i < n / 2; ++i) { int t = order[i]; order[i] = order[n - 1 - i]; order[n - 1 - i] = t; }
    int kept = 0;
    for (int pos = 0; pos < n; ++pos) {
        int p = order[pos];
        if (taken[p]) continue;
        taken[p] = 1;
        const int32_t* L = &lines[4 * p];
        int64_t A = L[3] - L[1], B = L[0] - L[2], C = (int64_t)L[2] * L[1] - (int64_t)L[3] * L[0];
        int64_t pcx = (L[2] + L[0]) >> 1, pcy = (L[3] + L[1]) >> 1; /* floor division by 2 */
        int64_t widest = 0;
        for (int qpos = pos; qpos < n; ++qpos) {
            int q = order[qpos];
            if (taken[q]) continue;
            const int32_t* M = &lines[4 * q];
            int64_t qcx = (M[2] + M[0]) >> 1, qcy = (M[3] + M[1]) >> 1;
            int64_t d2 = (pcx - qcx) * (pcx - qcx) + (pcy - qcy) * (pcy - qcy);
            if (d2 < len2[p] / 4) {
                taken[q] = 1;
                int64_t off = llabs(A * qcx + B * qcy + C);
                if (off > widest) widest = off;
            }
        }
        memcpy(&out_lines[4 * kept], L, 4 * sizeof(int32_t));
        double v = (double)widest / sqrt((double)(A * A + B * B)) / sqrt((double)len2[p]) * 2;
        nonline[kept] = v > 1 ? 1 : v;
        ++kept;
    }
    free(taken);
    free(order);
    free(len2);
    return kept;
}

/* ------------------------------------------------------------------ M3 detector (Detector.py:160-392) */
typedef struct {
    int H, W, n;
    int roi[4]; /* row0, col0, row1, col1 */
    int interval, adaptive, dy_on;
    double coef_a, coef_b;
    int hough_thr;
    double hough_min_len, hough_max_gap;
    double mask_area;
    /* state */
    int64_t t;
    uint8_t* ring;     /* n * HW */
    uint32_t* total;   /* HW */
    uint8_t* dy_ring;  /* n * HW */
    uint32_t* dy_total;
    int64_t dy_t;
    uint8_t *peak, *diff, *tmp0, *tmp1, *act, *dst;
    /* EMA (utils.py:334-368) */
    double ema_m0, ema_m, ema_v, ema_warm;
    int64_t ema_t;
    int bi_threshold;
    double bi_threshold_float, sigma;
} om3;

om3* om3_create(int H, int W, int n, const int* roi, int interval, int adaptive, int init_threshold, double coef_a,
                double coef_b, int hough_thr, double hough_min_len, double hough_max_gap, int dy_on, double mask_area) {
    om3* c = (om3*)calloc(1, sizeof(om3));
    size_t HW = (size_t)H * W;
    c->H = H; c->W = W; c->n = n;
    memcpy(c->roi, roi, sizeof(c->roi));
    c->interval = interval; c->adaptive = adaptive; c->dy_on = dy_on;
    c->coef_a = coef_a; c->coef_b = coef_b;
    c->hough_thr = hough_thr; c->hough_min_len = hough_min_len; c->hough_max_gap = hough_max_gap;
    c->mask_area = mask_area;
    c->ring = (uint8_t*)calloc((size_t)n * HW, 1);
    c->total = (uint32_t*)calloc(HW, sizeof(uint32_t));
    c->dy_ring = (uint8_t*)calloc((size_t)n * HW, 1);
    c->dy_total = (uint32_t*)calloc(HW, sizeof(uint32_t));
    c->peak = (uint8_t*)calloc(HW, 1); c->diff = (uint8_t*)calloc(HW, 1);
    c->tmp0 = (uint8_t*)calloc(HW, 1); c->tmp1 = (uint8_t*)calloc(HW, 1);
    c->act = (uint8_t*)calloc(HW, 1); c->dst = (uint8_t*)calloc(HW, 1);
    c->ema_m0 = 1 - (double)interval / 60; c->ema_m = c->ema_m0; c->ema_v = 0; c->ema_warm = n; c->ema_t = 0;
    c->bi_threshold = init_threshold; c->bi_threshold_float = init_threshold;
    return c;
}

void om3_destroy(om3* c) {
    if (!c) return;
    free(c->ring); free(c->total); free(c->dy_ring); free(c->dy_total);
    free(c->peak); free(c->diff); free(c->tmp0); free(c->tmp1); free(c->act); free(c->dst);
    free(c);
}

static void ema_update(om3* c, double x) {
    if (c->ema_warm != 0) {
        double ramp = c->ema_t * (1 - c->ema_m0) * c->ema_warm;
        if (ramp < 1) {
            double u = 1 - ramp;
            c->ema_m = c->ema_m0 * (1 - u * u);
        } else {
            c->ema_warm = 0;
            c->ema_m = c->ema_m0;
        }
    }
    c->ema_v = c->ema_m * c->ema_v + (1 - c->ema_m) * x;
    c->ema_t += 1;
}

/* LineDetector.update (Detector.py:225-229) incl. SNR_SW.update (Detector.py:73-91). */
void om3_update(om3* c, const uint8_t* frame) {
    const size_t HW = (size_t)c->H * c->W;
    const int n = c->n;
    c->t += 1;
    const int slot = (int)((c->t - 1) % n);
    uint8_t* cell = c->ring + (size_t)slot * HW;
    const int evict = c->t > n;
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < HW; ++p) {
        uint32_t s = c->total[p];
        if (evict) s -= cell[p];
        cell[p] = frame[p];
        c->total[p] = s + frame[p];
    }
    /* refresh_max: np.max over the whole ring (utils.py:302-307) */
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < HW; ++p) {
        uint8_t m = 0;
        for (int k = 0; k < n; ++k) {
            uint8_t v = c->ring[(size_t)k * HW + p];
            if (v > m) m = v;
        }
        c->peak[p] = m;
    }
    int k_used = 0;
    if (c->t > n && c->t % ((int64_t)c->interval * n) == 0) k_used = n;
    else if (c->t > 1 && c->t <= n) k_used = (int)c->t;
    if (k_used) {
        const int len = (int)(c->t < n ? c->t : n);
        int64_t s1 = 0, s2 = 0;
#pragma omp parallel for schedule(static) reduction(+ : s1, s2)
        for (int y = c->roi[0]; y < c->roi[2]; ++y)
            for (int x = c->roi[1]; x < c->roi[3]; ++x) {
                size_t p = (size_t)y * c->W + x;
                int m = (int)((c->total[p] / (uint32_t)len) & 0xFF);
                for (int k = 0; k < k_used; ++k) {
                    int d = (int)c->ring[(size_t)k * HW + p] - m;
                    s1 += d;
                    s2 += d * d;
                }
            }
        double cnt = (double)k_used * (c->roi[2] - c->roi[0]) * (c->roi[3] - c->roi[1]);
        double mu = (double)s1 / cnt;
        c->sigma = sqrt((double)s2 / cnt - mu * mu);
        ema_update(c, c->sigma);
    }
    if (c->adaptive && c->ema_v != 0) {
        c->bi_threshold_float = c->coef_a * (c->ema_v * c->ema_v) + c->coef_b;
        c->bi_threshold = (int)nearbyint(c->bi_threshold_float); /* Python round(): half to even */
    }
}

/* M3Detector.detect up to the raw Hough segments (Detector.py:324-360).  Returns total segment count. */
int om3_detect(om3* c, int32_t* lines, int cap, int* lit_count, double* dst_sum, double* gap_out) {
    const int H = c->H, W = c->W;
    const size_t HW = (size_t)H * W;
    const uint32_t len = (uint32_t)(c->t < c->n ? c->t : c->n);
    const int T = c->bi_threshold;
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < HW; ++p) c->diff[p] = (uint8_t)(c->peak[p] - (uint8_t)(c->total[p] / len));
    ocv_median3(c->diff, c->tmp0, H, W);
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < HW; ++p) c->tmp1[p] = c->tmp0[p] > T ? 255 : 0;
    ocv_dilate3(c->tmp1, c->tmp0, H, W);
    ocv_erode3(c->tmp0, c->act, H, W);
    if (c->dy_on) {
        /* calculate_dy_mask (Detector.py:234-242): second ring of binary images + running sum */
        c->dy_t += 1;
        const int slot = (int)((c->dy_t - 1) % c->n);
        uint8_t* cell = c->dy_ring + (size_t)slot * HW;
        const int evict = c->dy_t > c->n;
        const uint32_t dlen = (uint32_t)(c->dy_t < c->n ? c->dy_t : c->n);
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < HW; ++p) {
            uint32_t s = c->dy_total[p];
            if (evict) s -= cell[p];
            cell[p] = c->act[p];
            s += c->act[p];
            c->dy_total[p] = s;
            c->tmp0[p] = s <= (dlen - 1) * 255u;
        }
        ocv_erode3(c->tmp0, c->tmp1, H, W);
#pragma omp parallel for schedule(static)
        for (size_t p = 0; p < HW; ++p) c->dst[p] = (uint8_t)(c->act[p] * c->tmp1[p]);
    } else {
        memcpy(c->dst, c->act, HW);
    }
    int64_t lit = 0;
#pragma omp parallel for schedule(static) reduction(+ : lit)
    for (size_t p = 0; p < HW; ++p) lit += c->dst[p] == 255;
    *lit_count = (int)lit;
    *dst_sum = (double)lit / c->mask_area * 100;
    double gap = 1 - *dst_sum / 0.05;
    if (gap < 0) gap = 0;
    gap *= c->hough_max_gap;
    *gap_out = gap;
    return ocv_hough_lines_p(c->dst, H, W, c->hough_thr, c->hough_min_len, gap, lines, cap);
}

void om3_get(const om3* c, uint8_t* dst, uint8_t* act, int* bi_threshold, double* bi_threshold_float, double* snr,
             double* sigma) {
    const size_t HW = (size_t)c->H * c->W;
    if (dst) memcpy(dst, c->dst, HW);
    if (act) memcpy(act, c->act, HW);
    if (bi_threshold) *bi_threshold = c->bi_threshold;
    if (bi_threshold_float) *bi_threshold_float = c->bi_threshold_float;
    if (snr) *snr = c->ema_v;
    if (sigma) *sigma = c->sigma;
}

void om3_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int om3_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

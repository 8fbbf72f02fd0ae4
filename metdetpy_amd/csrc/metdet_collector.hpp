// metdet_collector.hpp -- native MeteorCollector bookkeeping (host code).
//
// What the reference does per frame with lines (collector.py:420-528) in Python objects and NumPy calls -- retire series
// that have been quiet for max_interval, chain the new responses to the active series or open new ones -- restated so that
// a whole pass of detector results is consumed in one call.  Every rule and quirk of the reference is kept:
//   * class ids come from an arg-max over the LINE axis and are zipped with the lines, so at most num_cls lines per frame
//     are used and "line i is meteor-like" means "the line maximising class i is line 0 or 1"   collector.py:486-495
//   * candidates for a new response are the series that were active before this frame plus the FIRST series opened in
//     this frame (the slice active[:num_active] shifts when a series is inserted before the sentinel)  collector.py:485, 507
//   * a series is judged once, max_interval after its last response: kept (-> waiting list), or dropped  collector.py:431-449
//   * the waiting list is exported when no active series could still merge with it                        collector.py:455-474
// Floating point follows NumPy's order of operations: np.std = sqrt(pairwise_sum((x - mean)^2) / n) with NumPy's pairwise
// summation (checked bit for bit against np.add.reduce / np.std in the tests); the library is built -ffp-contract=off.
// Output: finished series as plain records (unrounded doubles); rounding, time stamps, merging of the exported groups into
// MDRF records and the rescale to the raw resolution stay in Python (metdetpy_amd/collector.py), where the reference's own
// golden outputs pin them.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "../../include/metdet_hip.h"

namespace mdc {

// numpy/core/src/umath/loops_utils.h.src: DOUBLE_pairwise_sum (PW_BLOCKSIZE = 128)
static double pairwise_sum(const double* a, size_t n) {
    if (n < 8) {
        double res = -0.0;
        for (size_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        size_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    size_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

// np.std(x) (population): numpy/_core/_methods.py _var / _std
static double np_std(const std::vector<double>& x) {
    const size_t n = x.size();
    if (n == 0) return std::numeric_limits<double>::quiet_NaN();
    const double mean = pairwise_sum(x.data(), n) / (double)n;
    std::vector<double> d(n);
    for (size_t i = 0; i < n; ++i) {
        const double t = x[i] - mean;
        d[i] = t * t;
    }
    return std::sqrt(pairwise_sum(d.data(), n) / (double)n);
}

// np.sort on float64: ascending, NaNs last
static void np_sort(std::vector<double>& v) {
    std::sort(v.begin(), v.end(), [](double a, double b) {
        if (std::isnan(a)) return false;
        if (std::isnan(b)) return true;
        return a < b;
    });
}

// np.argmax: first maximum, a NaN counts as the maximum
static int np_argmax(const double* v, int n, int stride = 1) {
    int best = 0;
    for (int i = 1; i < n; ++i) {
        const double a = v[(size_t)i * stride], b = v[(size_t)best * stride];
        if (std::isnan(b)) break;
        if (std::isnan(a) || a > b) best = i;
    }
    return best;
}

struct ProbFunc {  // create_prob_func (collector.py:70-94)
    double a, b;
    // *is_int: the Python function returns the int literals 1 / 0 on its flat parts, and a product of four ints stays an
    // int, which shows in the emitted JSON ("score": 1 rather than 1.0)
    double operator()(double x, bool* is_int = nullptr) const {
        if (is_int) *is_int = false;
        if (x < a) return x / a;
        if (is_int) *is_int = true;
        if (a <= x && x <= b) return 1;
        if (is_int) *is_int = false;
        if (x < 2 * b) return (2 * b - x) / b;
        if (is_int) *is_int = true;
        return 0;
    }
};

struct Series {  // MeteorSeries (collector.py:141-369)
    std::vector<int64_t> px, py;       // coord_list.pts
    std::vector<int64_t> pframe;       // coord_list.frame_num
    std::vector<double> cx, cy;        // center_list.pts (float64 means)
    std::vector<double> drct;          // drct_list
    std::vector<double> cate_prob;
    int64_t start_frame = 0, end_frame = 0, last_activate_frame = 0;
    int64_t count = 1;
    int64_t rx0 = 65536, ry0 = 65536, rx1 = -65536, ry1 = -65536;  // range
};

static double pt_drct(int64_t x1, int64_t y1, int64_t x2, int64_t y2) {  // utils.py pt_drct: arccos((p2y - p1y) / |p2 - p1|)
    const double len = std::sqrt((double)((y1 - y2) * (y1 - y2) + (x1 - x2) * (x1 - x2)));
    return std::acos((double)(y2 - y1) / len);  // 0/0 -> NaN, as under np.errstate(invalid="ignore")
}

}  // namespace mdc

struct mdh_collector {
    mdh_collector_cfg cfg{};
    double max_interval = 0;       // m.max_interval * fps (float compare, collector.py:400)
    int64_t max_acti_frame = 0;    // int(m.max_interval * fps)
    double thre2 = 0;              // m.thre2 * exp_frame
    double runtime_length = 0;     // max(runtime_size)
    mdc::ProbFunc time_prob{}, speed_prob{}, len_prob{}, drct_prob{};
    std::vector<mdc::Series> active;   // without the reference's sentinel (it never matches, never retires, scores 0)
    std::vector<mdc::Series> waiting;
    std::vector<mdh_target> events;
    std::vector<double> centers;       // x, y pairs of the events' centre points
    int32_t group = 0;
    bool nan_cate_error = false;
    std::string err;

    double dist(const mdc::Series& s) const {
        return std::sqrt((double)((s.ry0 - s.ry1) * (s.ry0 - s.ry1) + (s.rx0 - s.rx1) * (s.rx0 - s.rx1)));
    }
    int cate(const mdc::Series& s) const { return mdc::np_argmax(s.cate_prob.data(), (int)s.cate_prob.size()); }
    double drst_std(const mdc::Series& s) const {  // collector.py:186-200
        if (s.drct.empty()) return 0;
        std::vector<double> d = s.drct;
        auto trimmed_std = [](std::vector<double> v) {
            if (v.size() >= 3) {
                mdc::np_sort(v);
                v.pop_back();
            }
            return mdc::np_std(v);
        };
        const double s1 = trimmed_std(d);
        for (double& v : d)
            if (v > M_PI / 2) v -= M_PI;
        const double s2 = trimmed_std(d);
        // Python's min(s1, s2): s2 if s2 < s1 else s1
        return s2 < s1 ? s2 : s1;
    }
    double prob_meteor(const mdc::Series& s, bool* is_int = nullptr) {  // collector.py:591-615
        const int c = cate(s);
        if (is_int) *is_int = false;
        if (c == 0) {
            const double duration = (double)(s.last_activate_frame - s.start_frame + 1);
            const double fix_duration = duration / cfg.fps;
            const double speed = dist(s) / ((double)(s.end_frame - s.start_frame) + 1e-6);
            const double fix_speed = speed * cfg.fps / runtime_length * 100;
            bool i1, i2, i3, i4;
            const double p = time_prob(fix_duration, &i1) * speed_prob(fix_speed, &i2) * len_prob(dist(s), &i3) * drct_prob(drst_std(s), &i4);
            if (is_int) *is_int = i1 && i2 && i3 && i4;
            return p;
        }
        for (double v : s.cate_prob)
            if (std::isnan(v)) {
                nan_cate_error = true;
                return std::numeric_limits<double>::quiet_NaN();
            }
        return s.cate_prob[c] / (double)s.count;
    }
    void emit(const mdc::Series& s, int flag) {  // MeteorSeries.get_met_attr + MeteorCollector.get_met_attr, unrounded
        mdh_target t;
        memset(&t, 0, sizeof(t));
        t.flag = flag;
        t.group = group;
        t.start_frame = s.start_frame;
        t.end_frame = s.end_frame;
        t.last_activate_frame = s.last_activate_frame;
        t.duration = s.last_activate_frame - s.start_frame + 1;
        const double d = dist(s);
        t.speed = d / ((double)(s.end_frame - s.start_frame) + 1e-6);
        // sort_range (collector.py:217-229): orient the span from the earliest to the latest response
        size_t e = 0, l = 0;
        for (size_t i = 1; i < s.pframe.size(); ++i) {
            if (s.pframe[i] < s.pframe[e]) e = i;
            if (s.pframe[i] > s.pframe[l]) l = i;
        }
        int64_t x0 = s.rx0, y0 = s.ry0, x1 = s.rx1, y1 = s.ry1;
        if (s.px[e] > s.px[l]) std::swap(x0, x1);
        if (s.py[e] > s.py[l]) std::swap(y0, y1);
        t.pt1[0] = (int32_t)x0; t.pt1[1] = (int32_t)y0; t.pt2[0] = (int32_t)x1; t.pt2[1] = (int32_t)y1;
        t.dist = std::sqrt((double)((y0 - y1) * (y0 - y1) + (x0 - x1) * (x0 - x1)));
        t.fix_dist = d / runtime_length * 100;
        t.fix_speed = t.speed * cfg.fps / runtime_length * 100;
        t.fix_motion_duration = (double)(s.end_frame - s.start_frame) / cfg.fps;
        t.fix_duration = (double)t.duration / cfg.fps;
        t.num_pts = (int32_t)s.px.size();
        t.category = cate(s);
        t.drct_loss = drst_std(s);
        bool score_int = false;
        t.score = prob_meteor(s, &score_int);
        t.score_is_int = score_int ? 1 : 0;
        t.center_off = (int32_t)(centers.size() / 2);
        t.n_center = (int32_t)s.cx.size();
        for (size_t i = 0; i < s.cx.size(); ++i) {
            centers.push_back(s.cx[i]);
            centers.push_back(s.cy[i]);
        }
        events.push_back(t);
    }
    void calc_new_range(mdc::Series& s, const int64_t (*pts)[2], int k) {
        for (int i = 0; i < k; ++i) {
            s.rx0 = std::min(s.rx0, pts[i][0]); s.ry0 = std::min(s.ry0, pts[i][1]);
            s.rx1 = std::max(s.rx1, pts[i][0]); s.ry1 = std::max(s.ry1, pts[i][1]);
        }
    }
    void append_points(mdc::Series& s, const int64_t (*pts)[2], int k, int64_t frame) {
        double sx = 0, sy = 0;
        for (int i = 0; i < k; ++i) {
            s.px.push_back(pts[i][0]); s.py.push_back(pts[i][1]); s.pframe.push_back(frame);
            sx += (double)pts[i][0]; sy += (double)pts[i][1];  // integer sums: exact
        }
        s.cx.push_back(sx / k);  // np.mean(pts, axis=0)
        s.cy.push_back(sy / k);
        calc_new_range(s, pts, k);
        s.drct.push_back(mdc::pt_drct(pts[0][0], pts[0][1], pts[1][0], pts[1][1]));
    }
    bool may_in_series(const mdc::Series& s, const int64_t (*pts)[2], int k, int64_t cur_frame) const {  // collector.py:357-369
        size_t first = s.pframe.size();
        for (size_t i = 0; i < s.pframe.size(); ++i)
            if (s.pframe[i] >= cur_frame - max_acti_frame) { first = i; break; }
        for (size_t i = first; i < s.pframe.size(); ++i)
            for (int j = 0; j < k; ++j) {
                const int64_t dy = pts[j][1] - s.py[i], dx = pts[j][0] - s.px[i];
                if ((double)(dy * dy + dx * dx) < thre2) return true;
            }
        return false;
    }

    // MeteorCollector.update (collector.py:420-528).  lines: n x 4, cates: n x num_cls (row-major).
    int update(int64_t cur_frame, const int32_t* lines, const double* cates, int n) {
        const int num_cls = cfg.num_cls;
        // 1. judge the series that have been quiet for max_interval
        std::vector<mdc::Series> keep, drop, still;
        for (mdc::Series& ms : active) {
            if ((double)(cur_frame - ms.last_activate_frame) >= max_interval) {
                const double p = prob_meteor(ms);
                if (nan_cate_error) { err = "nan detected in cate_prob"; return MDH_E_STATE; }
                const int c = cate(ms);
                if (p > cfg.det_thre / 2 && p != cfg.det_thre && !(c == num_cls - 2 || c == 1)) keep.push_back(std::move(ms));
                else drop.push_back(std::move(ms));
            } else {
                still.push_back(std::move(ms));
            }
        }
        active.swap(still);
        // DROP_FLAG export happens even when the list is empty in the reference (a no-op); groups only count real exports
        if (!drop.empty()) {
            for (const mdc::Series& ms : drop) emit(ms, 0);
            ++group;
        }
        for (mdc::Series& ms : keep) waiting.push_back(std::move(ms));
        // 2. export the waiting list when nothing active could still merge with it
        if (!waiting.empty()) {
            bool quiet = true;
            for (mdc::Series& ms : active) {
                if (prob_meteor(ms) > cfg.det_thre / 2 &&
                    (double)(ms.start_frame - waiting.back().last_activate_frame) <= max_interval) {
                    quiet = false;
                    break;
                }
            }
            if (nan_cate_error) { err = "nan detected in cate_prob"; return MDH_E_STATE; }
            if (quiet) {
                std::stable_sort(waiting.begin(), waiting.end(),
                                 [](const mdc::Series& a, const mdc::Series& b) { return a.start_frame < b.start_frame; });
                for (const mdc::Series& ms : waiting) emit(ms, 1);
                ++group;
                waiting.clear();
            }
        }
        if (n <= 0) return MDH_OK;
        // 3. chain the new responses
        const size_t n_old = active.size();
        bool opened = false;
        const int used = std::min(n, num_cls);  // zip(lines, cate_ids, cates): cate_ids has num_cls entries
        for (int i = 0; i < used; ++i) {
            const int cate_id = mdc::np_argmax(cates + i, n, num_cls);  // arg-max over the LINE axis of column i
            const int64_t x1 = lines[4 * i], y1 = lines[4 * i + 1], x2 = lines[4 * i + 2], y2 = lines[4 * i + 3];
            int64_t pts[5][2];
            int k;
            if (cate_id == 0 || cate_id == 1) {
                k = 3;
                pts[0][0] = x1; pts[0][1] = y1; pts[1][0] = x2; pts[1][1] = y2;
                pts[2][0] = floor_div2(x1 + x2); pts[2][1] = floor_div2(y1 + y2);
            } else {
                k = 5;
                pts[0][0] = x1; pts[0][1] = y1; pts[1][0] = x2; pts[1][1] = y2; pts[2][0] = x2; pts[2][1] = y1;
                pts[3][0] = x1; pts[3][1] = y2;
                pts[4][0] = (int64_t)((double)(x1 + x2) / 2); pts[4][1] = (int64_t)((double)(y1 + y2) / 2);  // int((x1+x2)/2)
            }
            const size_t ncand = std::min(active.size(), n_old + (opened ? 1 : 0));
            bool joined = false;
            for (size_t s = 0; s < ncand; ++s) {
                mdc::Series& ms = active[s];
                if (!may_in_series(ms, pts, k, cur_frame)) continue;
                // MeteorSeries.update (collector.py:319-346)
                for (int j = 0; j < k; ++j)
                    if (!(ms.rx0 <= pts[j][0] && pts[j][0] <= ms.rx1 && ms.ry0 <= pts[j][1] && pts[j][1] <= ms.ry1)) {
                        ms.end_frame = cur_frame;
                        break;
                    }
                ms.last_activate_frame = cur_frame;
                append_points(ms, pts, k, cur_frame);
                for (int c = 0; c < num_cls; ++c) ms.cate_prob[c] += cates[(size_t)i * num_cls + c];
                ms.count += 1;
                joined = true;
                break;
            }
            if (joined) continue;
            mdc::Series ns;
            ns.start_frame = std::max<int64_t>(cur_frame - 2 * cfg.exp_frame, 0);
            ns.end_frame = cur_frame;
            ns.last_activate_frame = cur_frame;
            ns.cate_prob.assign(cates + (size_t)i * num_cls, cates + (size_t)(i + 1) * num_cls);
            append_points(ns, pts, k, cur_frame);
            active.push_back(std::move(ns));
            opened = true;
        }
        return MDH_OK;
    }
    static int64_t floor_div2(int64_t v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }  // numpy // 2
};

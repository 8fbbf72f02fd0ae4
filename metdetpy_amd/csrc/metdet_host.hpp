// metdet_host.hpp -- the HOST-ONLY part of libmetdet_hip.so: the error text behind mdh_last_error, lineset_nms (utils.py:780-839)
// and the entry points of the native MeteorCollector (collector.py:372-631, metdet_collector.hpp).  No HIP anywhere in this file:
// metdet_hip.hip includes it, and metdet_host_only.cpp compiles it alone with g++ -fsanitize=address,undefined
// (`make -C metdetpy_amd/csrc san`: SURVEY section 5, "race detection / sanitizers"), driven by the collector's and the NMS's
// golden tests through MDH_LIB.
#pragma once
#include "metdet_collector.hpp"

#include "../../include/metdet_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

static thread_local std::string g_err;
static void set_err(const std::string& s) { g_err = s; }

extern "C" const char* mdh_last_error(void) { return g_err.c_str(); }
extern "C" int mdh_abi_version(void) { return MDH_ABI_VERSION; }

extern "C" int mdh_lineset_nms(const int32_t* lines, int n, int32_t* out_lines, double* nonline) {
    // utils.py:780-839.  Order: squared length descending; ties: later index first (stable ascending sort, reversed).
    if (n <= 0) return 0;
    std::vector<int64_t> len2(n);
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) {
        const int64_t dx = (int64_t)lines[4 * i + 2] - lines[4 * i], dy = (int64_t)lines[4 * i + 3] - lines[4 * i + 1];
        len2[i] = dx * dx + dy * dy;
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return len2[a] < len2[b]; });
    std::reverse(order.begin(), order.end());
    std::vector<char> taken(n, 0);
    int kept = 0;
    for (int pos = 0; pos < n; ++pos) {
        const int p = order[pos];
        if (taken[p]) continue;
        taken[p] = 1;
        const int32_t* L = lines + 4 * p;
        const int64_t A = (int64_t)L[3] - L[1], B = (int64_t)L[0] - L[2], C = (int64_t)L[2] * L[1] - (int64_t)L[3] * L[0];
        const int64_t pcx = ((int64_t)L[2] + L[0]) >> 1, pcy = ((int64_t)L[3] + L[1]) >> 1;  // floor //2
        int64_t widest = 0;
        for (int qp = pos; qp < n; ++qp) {
            const int q = order[qp];
            if (taken[q]) continue;
            const int32_t* M = lines + 4 * q;
            const int64_t qcx = ((int64_t)M[2] + M[0]) >> 1, qcy = ((int64_t)M[3] + M[1]) >> 1;
            const int64_t d2 = (pcx - qcx) * (pcx - qcx) + (pcy - qcy) * (pcy - qcy);
            if (d2 < len2[p] / 4) {
                taken[q] = 1;
                widest = std::max<int64_t>(widest, std::llabs(A * qcx + B * qcy + C));
            }
        }
        if (out_lines) memcpy(out_lines + 4 * kept, L, 4 * sizeof(int32_t));
        if (nonline) {
            const double v = (double)widest / std::sqrt((double)(A * A + B * B)) / std::sqrt((double)len2[p]) * 2;
            nonline[kept] = v > 1 ? 1 : v;
        }
        ++kept;
    }
    return kept;
}


// ------------------------------------------------------------------------------------------------ native collector
extern "C" mdh_collector* mdh_collector_create(const mdh_collector_cfg* cfg) {
    if (!cfg) { set_err("mdh_collector_create: null cfg"); return nullptr; }
    if (cfg->struct_size != sizeof(mdh_collector_cfg)) {
        set_err("mdh_collector_create: cfg.struct_size is " + std::to_string(cfg->struct_size) + ", this library's mdh_collector_cfg has " +
                std::to_string(sizeof(mdh_collector_cfg)) + " bytes: the caller was built against another include/metdet_hip.h");
        return nullptr;
    }
    if (!(cfg->fps > 0) || cfg->num_cls < 2 || cfg->exp_frame < 1 || cfg->runtime_w < 1 || cfg->runtime_h < 1) {
        set_err("mdh_collector_create: bad configuration");
        return nullptr;
    }
    mdh_collector* c = new mdh_collector();
    c->cfg = *cfg;
    c->max_interval = cfg->max_interval * cfg->fps;
    c->max_acti_frame = (int64_t)(cfg->max_interval * cfg->fps);  // int(): truncation
    c->thre2 = cfg->thre2 * cfg->exp_frame;
    c->runtime_length = (double)std::max(cfg->runtime_w, cfg->runtime_h);
    c->time_prob = mdc::ProbFunc{cfg->time_range[0], cfg->time_range[1]};
    c->speed_prob = mdc::ProbFunc{cfg->speed_range[0], cfg->speed_range[1]};
    c->len_prob = mdc::ProbFunc{cfg->min_len, std::numeric_limits<double>::infinity()};
    c->drct_prob = mdc::ProbFunc{cfg->drct_range[0], cfg->drct_range[1]};
    return c;
}
extern "C" void mdh_collector_destroy(mdh_collector* c) { delete c; }
extern "C" int mdh_collector_active(const mdh_collector* c) { return c ? (int)c->active.size() : -1; }

extern "C" int mdh_collector_update(mdh_collector* c, int64_t cur_frame, const int32_t* lines, const double* cates, int n) {
    if (!c || (n > 0 && (!lines || !cates))) { set_err("mdh_collector_update: null argument"); return MDH_E_ARG; }
    const int rc = c->update(cur_frame, lines, cates, n);
    if (rc != MDH_OK) set_err("mdh_collector_update: " + c->err);
    return rc;
}

extern "C" int mdh_collector_update_batch(mdh_collector* c, int K, int64_t frame0, int64_t it0, int eq_int_fps,
                                          const mdh_frame_info* info, const int32_t* nms_lines, const double* nonline) {
    if (!c || !info || !nms_lines || !nonline || K < 0 || eq_int_fps < 1) { set_err("mdh_collector_update_batch: bad argument"); return MDH_E_ARG; }
    const int num_cls = c->cfg.num_cls;
    std::vector<double> cates;
    for (int i = 0; i < K; ++i) {
        const int n = (info[i].n_raw > 0 && info[i].n_raw <= 500) ? info[i].n_nms : 0;  // NUM_LINES_TOOMUCH (Detector.py:30, 358-360)
        if (n == 0 && (it0 + i) % eq_int_fps != 0) continue;
        cates.assign((size_t)n * num_cls, 0.0);
        const double* p = nonline + (size_t)i * MDH_MAX_LINES;
        for (int k = 0; k < n; ++k) {
            cates[(size_t)k * num_cls + num_cls - 1] = p[k];
            cates[(size_t)k * num_cls] = 1 - p[k];
        }
        const int rc = c->update(frame0 + (int64_t)i * c->cfg.exp_frame, nms_lines + (size_t)i * MDH_MAX_LINES * 4, cates.data(), n);
        if (rc != MDH_OK) { set_err("mdh_collector_update_batch: " + c->err); return rc; }
    }
    return MDH_OK;
}

extern "C" int mdh_collector_events(mdh_collector* c, mdh_target* out, int cap, double* centers, int cap_centers, int* n_centers) {
    if (!c) { set_err("mdh_collector_events: null collector"); return MDH_E_ARG; }
    const int n = (int)c->events.size(), nc = (int)(c->centers.size() / 2);
    if (n_centers) *n_centers = nc;
    if (n <= cap && nc <= cap_centers && out && (centers || nc == 0)) {
        if (n) memcpy(out, c->events.data(), sizeof(mdh_target) * n);
        if (nc) memcpy(centers, c->centers.data(), sizeof(double) * 2 * nc);
        c->events.clear();
        c->centers.clear();
    }
    return n;
}

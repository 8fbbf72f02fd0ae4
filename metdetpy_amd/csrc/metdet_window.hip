// metdet_window.hip -- the wide-load window kernels of ONE part of the window lengths (-DMDH_WIN_PART=0..5, see metdet_window.hpp):
// k_stack_threshold_wide<N, 16 | 8, per-frame, dense> and k_stack_threshold_mail<N, 16 | 8> for the part's N, behind three plain
// functions the main translation unit calls.  gfx950 only.
#define MDH_WINDOW_ONLY 1  // metdet_kernels.hpp: the window kernels and what they need, nothing else
#include "metdet_window.hpp"

#ifndef MDH_WIN_PART
#error "compile with -DMDH_WIN_PART=0..5"
#endif
#define WIN_CAT2(a, b) a##b
#define WIN_CAT(a, b) WIN_CAT2(a, b)
#define WIN_FN(name) WIN_CAT(name, MDH_WIN_PART)

namespace {
constexpr int kRanges[MDH_WIN_PARTS][2] = MDH_WIN_PART_RANGES;
constexpr int kFirst = kRanges[MDH_WIN_PART][0], kLast = kRanges[MDH_WIN_PART][1];

template <int N, int PPL>
int launch_one(const WindowLaunch& w) {
    if constexpr (PPL == 16 && N > 16) {
        return 0;
    } else {
        switch (w.kind) {
            case WIN_LEAN:
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stack_threshold_wide<N, PPL, true, false>), w.grid, w.block, w.lds, w.stream, w.ptrtab, w.thr, w.bbits,
                                   w.blockcnt, w.g, w.Tc, w.lean, w.nzwords, w.wrec);
                return 1;
            case WIN_DENSE:
                if constexpr (PPL == 16 || N > 16) {
                    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stack_threshold_wide<N, PPL, false, true>), w.grid, w.block, w.lds, w.stream, w.ptrtab, w.thr,
                                       w.bbits, w.blockcnt, w.g, w.Tc, w.lean, w.nzwords, w.wrec);
                    return 1;
                }
                return 0;
            default:
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stack_threshold_wide<N, PPL, false, false>), w.grid, w.block, w.lds, w.stream, w.ptrtab, w.thr, w.bbits,
                                   w.blockcnt, w.g, w.Tc, w.lean, w.nzwords, w.wrec);
                return 1;
        }
    }
}
template <int N, int PPL>
int mail_one(const MailLaunch& m) {
    if constexpr ((PPL == 16) != (N <= 16)) {  // (launched ahead with 16 pixels per lane up to n = 16, with 8 above: mail_ppl)
        return 0;
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_stack_threshold_mail<N, PPL>), dim3(m.grid, 1), dim3(256), 0, m.stream, m.mail, m.seq, m.go, m.cl);
        return 1;
    }
}
template <int N, int PPL>
int occupancy_one() {
    if constexpr (PPL == 16 && N > 16) {
        return 0;
    } else {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_stack_threshold_wide<N, PPL, false, false>, 256, 0) != hipSuccess || nb < 1) {
            (void)hipGetLastError();
            nb = 2;  // (what the largest instantiations run at)
        }
        return nb < 8 ? nb : 8;
    }
}

// N = kFirst + I for I = 0 .. kLast - kFirst: a compile-time walk over the part's window lengths
template <int N>
int launch_n(const WindowLaunch& w) {
    if (w.n == N) return w.ppl == 16 ? launch_one<N, 16>(w) : launch_one<N, 8>(w);
    if constexpr (N < kLast) return launch_n<N + 1>(w);
    return 0;
}
template <int N>
int mail_n(const MailLaunch& m) {
    if (m.n == N) return m.ppl == 16 ? mail_one<N, 16>(m) : mail_one<N, 8>(m);
    if constexpr (N < kLast) return mail_n<N + 1>(m);
    return 0;
}
template <int N>
int occupancy_n(int n, int ppl) {
    if (n == N) return ppl == 16 ? occupancy_one<N, 16>() : occupancy_one<N, 8>();
    if constexpr (N < kLast) return occupancy_n<N + 1>(n, ppl);
    return 0;
}
}  // namespace

int WIN_FN(mdh_win_launch_p)(const WindowLaunch& w) { return (w.n < kFirst || w.n > kLast || !wide_exists(w.n, w.ppl)) ? 0 : launch_n<kFirst>(w); }
int WIN_FN(mdh_win_mail_p)(const MailLaunch& m) { return (m.n < kFirst || m.n > kLast || !wide_exists(m.n, m.ppl)) ? 0 : mail_n<kFirst>(m); }
int WIN_FN(mdh_win_occupancy_p)(int n, int ppl) { return (n < kFirst || n > kLast || !wide_exists(n, ppl)) ? 0 : occupancy_n<kFirst>(n, ppl); }

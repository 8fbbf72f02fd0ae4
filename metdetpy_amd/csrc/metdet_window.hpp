// metdet_window.hpp -- hand-over between libmetdet_hip's main translation unit (metdet_hip.hip) and the translation units that hold
// the wide-load window kernels (metdet_window.hip, compiled once per MDH_WIN_PART).  Round 6: the window lengths 17 .. 32 brought the
// number of instantiations of k_stack_threshold_wide / _mail to ~190, each a fully unrolled window period: in one translation unit the
// library took five minutes to build; as seven objects built side by side (make -j) it takes about one.
#pragma once
#include "metdet_kernels.hpp"

enum { WIN_SPARSE = 0, WIN_DENSE = 1, WIN_LEAN = 2 };  // WindowLaunch.kind: the batched sparse / dense output formats, the per-frame instantiation

struct WindowLaunch {  // one launch of k_stack_threshold_wide<n, ppl, kind == WIN_LEAN, kind == WIN_DENSE>
    int n, ppl, kind;
    dim3 grid, block;
    unsigned lds;
    hipStream_t stream;
    const uint8_t* const* ptrtab;
    const int* thr;
    ActWord* bbits;
    unsigned* blockcnt;
    PassGeom g;
    int Tc;
    LeanArgs lean;
    unsigned long long* nzwords;
    const WinRec* wrec;
};
struct MailLaunch {  // one launch of k_stack_threshold_mail<n, ppl>
    int n, ppl;
    unsigned grid;
    hipStream_t stream;
    MailBox* mail;
    unsigned seq;
    unsigned* go;
    MailClean cl;
};

// which wide-load instantiations exist: 16 pixels per lane for n = 2 .. 16, 8 per lane for n = 2 .. 32; a dense output format for
// the 16-pixel ones and for the 8-pixel ones of the long windows (n > 16)
inline bool wide_exists(int n, int ppl) { return ppl == 16 ? (n >= 2 && n <= 16) : (ppl == 8 && n >= 2 && n <= 32); }
inline bool wide_has_dense(int n, int ppl) { return ppl == 16 ? (n >= 2 && n <= 16) : (ppl == 8 && n > 16 && n <= 32); }

// The parts: window lengths [first, last] each translation unit instantiates.  Every entry point returns 0 when the window
// length is another part's, otherwise: *_launch 1 (launched; errors through hipGetLastError as for any launch), *_occupancy the
// resident 256-thread workgroups per CU of the sparse instantiation (>= 1).
#define MDH_WIN_PARTS 6
#define MDH_WIN_PART_RANGES {{2, 8}, {9, 12}, {13, 16}, {17, 22}, {23, 27}, {28, 32}}
#define MDH_WIN_DECL(P)                                   \
    int mdh_win_launch_p##P(const WindowLaunch& w);       \
    int mdh_win_mail_p##P(const MailLaunch& m);           \
    int mdh_win_occupancy_p##P(int n, int ppl);
MDH_WIN_DECL(0) MDH_WIN_DECL(1) MDH_WIN_DECL(2) MDH_WIN_DECL(3) MDH_WIN_DECL(4) MDH_WIN_DECL(5)
#undef MDH_WIN_DECL

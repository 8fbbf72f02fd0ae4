#!/usr/bin/env python3
"""Reproducer hunt for the abort inside mdh_pre_run (2 of round 5's 10 full GPU sessions ended with SIGABRT there).

One long-lived process that has run the detector paths first, then ITERS iterations of host-fed preprocessing from PAGEABLE source
arrays that alternate between large (several MB: the runtime pins such memory on the fly) and small (below a megabyte: the runtime's
staging path), freed and reallocated every iteration so that glibc hands the same addresses out again (large arrays are mmap'd and
munmap'd: the address of a freed 6 MB array is the next one's), on preprocessors (streams) created a moment before, with detector
contexts created and destroyed in between -- what tests/test_gpu_preprocess.py does minutes into a session of the whole suite.

    MDH_PRE_STAGE=0 python tools/stress_pre_run.py --iters 2000      the old path: hipMemcpyAsync straight from the caller's memory
    MDH_PRE_STAGE=1 python tools/stress_pre_run.py --iters 2000      the library's own pinned staging buffers (round 5)
    --mode detector   the same hunt on the detector's host-fed paths (mdh_process_batch from pageable frames, mdh_update from fresh arrays)
    --mode stale      the hypothesis test: a frame buffer the library registered on a loader's behalf (mdh_set_auto_register) is FREED
                      by its owner WITHOUT mdh_host_unregister (against the documented order), and fresh arrays of several sizes -- which
                      the allocator places on the freed range -- are then handed to the preprocessing
    --mode stale-detector   the same, the fresh arrays handed to the detector's host-fed paths (mdh_process_batch, mdh_update)

Prints one JSON line at the end (iterations done, mismatches); an abort leaves the runtime's last words on stderr and a non-zero
exit code instead (run under `timeout`, keep stderr).
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2000)
    ap.add_argument("--mode", default="pre", choices=["pre", "detector", "stale", "stale-detector"])
    ap.add_argument("--seconds", type=float, default=600.0, help="stop after this many seconds even if --iters has not been reached")
    a = ap.parse_args()
    from metdetpy_amd import M3Detector
    from metdetpy_amd.imgproc import Preprocessor, max_stack
    from metdetpy_amd.synth import SkyStream
    from oracle import m3_oracle as mo
    from tests.helpers import binary_cfg

    rng = np.random.default_rng(5)
    # the detector paths first: batched passes from pageable memory, the per-frame protocol from fresh arrays and from a pinned pool
    t0 = time.time()
    shape = (270, 480)
    mask = np.ones(shape, np.uint8)
    frames = SkyStream(*shape, 7, speed=5, streak_period=11, streak_life=7).take(64)
    for n in (4, 8, 16, 25):
        det = M3Detector(1, n, mask, 10, binary_cfg("normal"), None, max_batch=32)
        det.detect_batch(frames)
        for f in frames[:24]:
            det.update(f.copy())
            det.detect()
        det.close()
    done = bad = 0
    big_shapes = [(4, 540, 960, 3), (2, 1080, 1920, 3), (1, 432, 768, 3), (3, 540, 960, 3)]
    small_shapes = [(1, 67, 131, 3), (2, 120, 200, 3), (1, 216, 384, 3)]
    keep = []  # a few arrays kept alive for a while and then dropped: holes of every size in the heap and in the mmap region
    for it in range(a.iters):
        if time.time() - t0 > a.seconds:
            break
        shp = big_shapes[(it // 2) % len(big_shapes)] if it % 2 == 0 else small_shapes[(it // 2) % len(small_shapes)]
        if a.mode == "pre":
            src = rng.integers(0, 256, shp, dtype=np.uint8)  # fresh pageable memory (freed at the end of the iteration)
            dst_wh = (shp[2] // 2, shp[1] // 2) if it % 3 else (shp[2], shp[1])
            pre = Preprocessor(shp[1:3], dst_wh, None)  # (a stream created a moment before the copy)
            got = pre(src)
            if it % 97 == 0:
                want = mo.merge_max([mo.preprocess_frame(f, dst_wh, np.ones((dst_wh[1], dst_wh[0]), np.uint8)) for f in src])
                bad += int(not np.array_equal(got, want))
            pre.close()
            if it % 5 == 0:
                max_stack(src.reshape(shp[0], -1))
            if it % 7 == 0:
                keep.append(src)
            if len(keep) > 6:
                del keep[:3]
            del src, got
        elif a.mode in ("stale", "stale-detector"):
            import ctypes
            hw = [(1080, 1920), (540, 960), (720, 1280)][it % 3]
            m1 = np.ones(hw, np.uint8)
            det = M3Detector(1, 4, m1, 10, binary_cfg("normal"), None, max_batch=8)
            det._ctx.set_auto_register(True)
            buf = np.empty(hw, np.uint8)  # >= 512 KB: mmap'd by the allocator
            for k in range(8):
                buf[...] = 30 + (k & 1)
                det.update(buf)
                det.detect()
            reg = det._ctx.update_stats()["auto_registered"]
            del buf  # freed (munmap'd) while registered: the loader broke the contract
            if a.mode == "stale":  # the preprocessing from fresh arrays that land on the freed range (what aborted in round 5's sessions)
                for rep_ in range(3):
                    src = rng.integers(0, 256, (1 + (it + rep_) % 3, hw[0] // 2 if rep_ % 2 else hw[0], hw[1], 3), dtype=np.uint8)
                    pre = Preprocessor(src.shape[1:3], (src.shape[2] // 2, src.shape[1] // 2), None)
                    got = pre(src)
                    pre.close()
                    del src, got
            else:              # the detector's host-fed paths (they hand pageable memory to the runtime's own copy)
                fr = rng.integers(20, 40, (6,) + hw, dtype=np.uint8)
                det.detect_batch(fr)
                det.update(fr[0].copy())
                det.detect()
                del fr
            det.close()
            bad += int(reg != 1)
        else:
            n = (4, 8, 16)[it % 3]
            K = 8 + (it % 5) * 8
            fr = rng.integers(20, 40, (K,) + shape, dtype=np.uint8)  # fresh pageable frames
            det = M3Detector(1, n, mask, 10, binary_cfg("normal"), None, max_batch=16)
            det.detect_batch(fr)
            for k in range(6):
                det.update(fr[k].copy())
                det.detect()
            det.close()
            del fr
        if it % 50 == 49:
            det = M3Detector(1, 8, mask, 10, binary_cfg("normal"), None, max_batch=32)
            det.detect_batch(frames)
            det.close()
            gc.collect()
        done += 1
    print(json.dumps({"mode": a.mode, "MDH_PRE_STAGE": os.environ.get("MDH_PRE_STAGE", "1 (default)"), "iterations": done, "mismatches": bad,
                      "seconds": round(time.time() - t0, 1)}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- frames/sec of the M3 line-detector hot path on synthetic 1920x1080 uint8 streams (BASELINE config C2).

    python bench.py --gpus N --steps K --warmup W

One "step" = one native pass (mdh_submit_batch / mdh_collect_batch, several passes in flight) over a batch of `--batch` consecutive frames of a seeded synthetic
night-sky stream that is already resident in HBM (a pool of `--batch` frames per GPU, cycled): update + detect incl.
Hough and NMS, results on the host.  N > 1: one process per GPU (torchrun), one independent stream per GPU
(BASELINE config C4), no data-path collective; RCCL only gathers the per-rank detection counts at the end.
`value` = frames all ranks processed / max-over-ranks time.

Also on the JSON line:
  roofline      dominant pixel kernel (stack_threshold): algorithmic HBM reads per launch ((n+1)*H*W per frame,
                SURVEY.md 8d) / mean launch duration measured with HIP events on the library's own stream.
  cpu_baseline  the C port of the CPU oracle (oracle/metdet_oracle.c, OpenMP) timed on this box's host cores over a
                bounded sample of the same stream (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the library's streams need their own hardware queues (the HIP default is 4); must be set before HIP / torch start
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak

WORKLOADS = {
    # name: (H, W, n, seed, use_mask)
    "C2": dict(H=1080, W=1920, n=8, seed=20220413, desc="synthetic 1920x1080 u8 stream, 8-frame window"),
    "C3": dict(H=2160, W=3840, n=16, seed=20220414, desc="synthetic 3840x2160 u8 stream, 16-frame window + mask"),
    "C1s": dict(H=540, W=960, n=8, seed=20220412, desc="synthetic 960x540 u8 stream (runtime size of config 1)"),
    "C5": dict(H=4320, W=7680, n=8, seed=20220415, sensitivity="high",
               desc="synthetic 7680x4320 u8 stream, m3det_high thresholds, whole frame on one GPU (no row-band tiling)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--batch", type=int, default=2048, help="frames per step (and frames resident in HBM per GPU)")
    ap.add_argument("--unique", type=int, default=256, help="distinct synthetic frames generated per GPU; the resident "
                    "batch repeats them (generation is NumPy on the host and would otherwise dominate the run time)")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--parity-frames", type=int, default=256, help="frames of the stream compared against the CPU oracle "
                    "after the timed region (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--host-input", action="store_true", help="frames come from PINNED HOST memory every pass (PCIe-inclusive rate; "
                    "never the headline value, see DESIGN.md)")
    return ap.parse_args()


SENSITIVITY = "normal"


def make_cfg():
    from metdetpy_amd import BinaryCfg, BinaryCoreCfg, DynamicCfg, HoughLineCfg
    # config/m3det_normal.json (or m3det_high.json) detector block
    return BinaryCfg(BinaryCoreCfg(True, 7, SENSITIVITY, 0.1, 2), HoughLineCfg(10, 10, 10), DynamicCfg(True, 5))


def load_mask(H, W):
    """test/mask-east.jpg binarised by the loader rule (fixture committed as tests/golden/mask_east_*.npy)."""
    path = os.path.join(ROOT, "tests", "golden", f"mask_east_{W}x{H}.npz")
    if os.path.exists(path):
        return np.load(path)["mask"]
    return None


def cpu_baseline(frames, n, mask, seconds):
    """Time the C port of the oracle (update + detect up to raw Hough + NMS) on the host cores."""
    from oracle import c_oracle as co
    from metdetpy_amd.utils import select_std_roi
    roi = select_std_roi(mask, 0.1)
    # the port scales to ~16 threads on the GPU box's host (83 fps at 1080p; 32 threads are slower: the frame-sized
    # loops are memory-bound and the PPHT is serial)
    co.lib().om3_set_threads(min(16, len(os.sched_getaffinity(0))))
    det = co.CM3(n, mask, roi, sensitivity=SENSITIVITY)
    threads = int(co.lib().om3_num_threads())
    warm = min(2 * n, len(frames))
    for i in range(warm):
        det.update(frames[i])
        det.detect()
    done, t0 = 0, time.perf_counter()
    while True:
        f = frames[(warm + done) % len(frames)]
        det.update(f)
        raw = det.detect()
        if 0 < len(raw) <= 500:
            co.lineset_nms(raw)
        done += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or done >= 4 * len(frames):
            break
    return dict(value=done / dt, unit="frames/s", cores=threads, kind="port",
                sample=f"{done} frames of the same stream after {warm} warm-up frames, oracle/metdet_oracle.c "
                       f"(gcc -O3 -fopenmp, {threads} threads), host cpu_count={os.cpu_count()}")


def parity_check(frames, n, mask, local, count):
    """Detection parity vs the CPU reference restatement (the metric's second half): a fresh GPU detector and the C
    port of the oracle see the first `count` frames of the stream; thresholds, lit counts, raw Hough segments (in
    discovery order) and NMS output must be identical, and so must the records the collector makes of them."""
    from metdetpy_amd import M3Detector
    from metdetpy_amd.utils import select_std_roi
    from oracle import c_oracle as co
    count = min(count, len(frames))
    det = M3Detector(1, n, mask, 10, make_cfg(), None, device=local, max_batch=count)
    info, raw, nms, prob = det._ctx.process_batch(np.ascontiguousarray(frames[:count]))
    det.close()
    co.lib().om3_set_threads(min(16, len(os.sched_getaffinity(0))))
    orc = co.CM3(n, mask, select_std_roi(mask, 0.1), sensitivity=SENSITIVITY)
    bad, segs, o_results = [], 0, []
    for i in range(count):
        orc.update(frames[i])
        o_raw = orc.detect()
        o_results.append((np.zeros((0, 4), np.int32), np.zeros(0)))
        thr = orc.state()[0]
        ok = (int(info[i]["bi_threshold"]) == thr and int(info[i]["lit_count"]) == orc.lit_count
              and int(info[i]["n_raw"]) == orc.lines_num
              and np.array_equal(raw[i][:min(orc.lines_num, 512)], o_raw[:512]))
        if 0 < orc.lines_num <= 500:
            o_nms, o_prob = co.lineset_nms(o_raw)
            o_results[i] = (o_nms, o_prob)
            k = int(info[i]["n_nms"])
            ok = ok and k == len(o_nms) and np.array_equal(nms[i][:k], o_nms) and np.array_equal(prob[i][:k], o_prob)
        segs += orc.lines_num
        if not ok:
            bad.append(i)
    # the records a collector makes of them (config/m3det_normal.json's meteor_cfg, re-check off): GPU results through the
    # native collector one pass at a time, oracle results through the Python collector frame by frame
    from metdetpy_amd.collector import MeteorCollector, NativeMeteorCollector
    from metdetpy_amd.metstruct import CollectorCfg, MeteorCfg, RecheckSwitch, RuntimeParams
    H, W = mask.shape
    ccfg = CollectorCfg(MeteorCfg(15, 4, [0, 8], [2, 21], [0, 0.6], 0.5, 2048), RecheckSwitch(False), {})
    rt = RuntimeParams(fps=float(n), exp_frame=1, eq_fps=float(n), eq_int_fps=n, exp_time=1.0 / n, runtime_size=[W, H],
                       raw_size=[W, H], positive_category_list=["METEOR", "RED_SPRITE"])
    gcol, ocol = NativeMeteorCollector(ccfg, rt), MeteorCollector(ccfg, rt)
    gcol.update_batch(0, 0, info, nms, prob)
    gcol.clear()
    for i in range(count):
        k = len(o_results[i][0])
        if k or i % rt.eq_int_fps == 0:
            cates = np.zeros((k, 10))
            cates[:, -1] = o_results[i][1]
            cates[:, 0] = 1 - o_results[i][1]
            ocol.update(i, o_results[i][0], cates)
    ocol.clear()
    g_rec, o_rec = [r.to_dict() for r in gcol.met_exporter.meteor_list], [r.to_dict() for r in ocol.met_exporter.meteor_list]
    rec_ok = g_rec == o_rec
    return {"frames_checked": count, "raw_segments_checked": segs, "mismatching_frames": bad, "records": len(o_rec),
            "records_identical": rec_ok, "verdict": "identical" if not bad and rec_ok else "MISMATCH",
            "checked": "bi_threshold, lit_count, raw Hough segments (order included), lineset_nms output, collector records "
                       "vs oracle/metdet_oracle.c"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    tdev = None
    if world > 1:
        import torch
        import torch.distributed as dist
        # "nccl" is RCCL on ROCm.  MDH_BENCH_BACKEND=gloo (CPU tensors) exists to exercise the multi-rank path on a box with
        # fewer GPUs than ranks: ranks then share devices round-robin.
        backend = os.environ.get("MDH_BENCH_BACKEND", "nccl")
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            tdev = torch.device("cuda", local)
            dist.init_process_group("nccl", device_id=tdev)
        else:
            tdev = torch.device("cpu")
            dist.init_process_group(backend)
    wl = WORKLOADS[a.workload]
    H, W, n = wl["H"], wl["W"], wl["n"]
    global SENSITIVITY
    SENSITIVITY = wl.get("sensitivity", "normal")
    from metdetpy_amd import M3Detector
    from metdetpy_amd import _lib
    from metdetpy_amd.synth import SkyStream

    mask = np.ones((H, W), np.uint8)
    if a.workload == "C3":
        m = load_mask(H, W)
        if m is not None:
            mask = m
    B = a.batch
    stream = SkyStream(H, W, wl["seed"] + rank, mask01=None if mask.all() else mask)
    uniq = min(B, max(a.unique, 2 * n))
    frames = stream.take(uniq)
    det = None
    while det is None:  # the buffers of a 2048-frame pass take ~130 GB at 1080p: halve the pass if the GPU cannot give that
        try:
            det = M3Detector(1, n, mask, 10, make_cfg(), None, device=local, max_batch=B)
        except _lib.MetDetHipError as e:
            if B <= 64 or "hipMalloc" not in str(e):
                raise
            print(f"[bench] {e}; retrying with {B // 2} frames per pass", file=sys.stderr, flush=True)
            B //= 2
            uniq = min(uniq, B)
            frames = frames[:uniq]
    ctx = det._ctx
    if a.host_input:
        import ctypes
        pool = ctx.host_alloc(B * H * W)
        for start in range(0, B, uniq):
            cnt = min(uniq, B - start)
            ctypes.memmove(pool + start * H * W, frames[:cnt].ctypes.data, cnt * H * W)
    else:
        pool = ctx.device_alloc(B * H * W)
        for start in range(0, B, uniq):  # resident batch = the unique frames repeated
            cnt = min(uniq, B - start)
            ctx.device_upload(pool + start * H * W, frames[:cnt])
    info = np.zeros(B, _lib.FRAME_INFO_DTYPE)
    raw = np.zeros((B, _lib.MDH_MAX_LINES, 4), np.int32)
    nms = np.zeros((B, _lib.MDH_MAX_LINES, 4), np.int32)
    prob = np.zeros((B, _lib.MDH_MAX_LINES), np.float64)

    tally = [0, 0, 0]

    def collect():
        ctx.collect_into(info, raw, nms, prob)
        tally[0] += int(info["n_raw"].sum())
        tally[1] += int(info["n_nms"].sum())
        tally[2] += int(info["lit_count"].sum())

    def run(nsteps):
        """nsteps passes, up to ctx.max_in_flight queued: later passes are submitted before earlier ones are collected,
        so the pixel stages of one pass overlap the Hough stage of the previous one and the host stays ahead of the
        GPU.  Every pass is fully collected before run() returns."""
        for _s in range(nsteps):
            if ctx.in_flight == ctx.max_in_flight:
                collect()
            ctx.submit(pool, B, not a.host_input)
        while ctx.in_flight:
            collect()

    def fence():
        # the library has no work in flight between run() calls; the barrier aligns the ranks
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    if a.warmup:
        run(a.warmup)
    if not a.no_profile:
        ctx.profile_enable(4)  # every 4th pass carries timing events (they cost the pipeline a few per cent when on every pass)
        ctx.profile_read(reset=True)
    tally[:] = [0, 0, 0]
    fence()
    t0 = time.perf_counter()
    run(a.steps)
    fence()
    segs, kept, lit = tally
    dt = time.perf_counter() - t0
    prof = ctx.profile_read(reset=True) if not a.no_profile else {}
    ctx.profile_enable(False)
    if dist is not None:
        import torch
        tmax = torch.tensor([dt], device=tdev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_all = float(tmax.item())
        counts = torch.tensor([segs, kept, lit], device=tdev, dtype=torch.int64)
        gathered = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(gathered, counts)  # the final detection gather (RCCL)
        tot = torch.stack(gathered).sum(0).tolist()
    else:
        dt_all = dt
        tot = [segs, kept, lit]
    frames_total = a.steps * B * world
    fps = frames_total / dt_all
    out = {
        "metric": "frames/sec at 1920x1080 (1/2/4/8 GPU) + detection parity vs CPU ref",
        "value": fps,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": dt_all / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{a.workload}: {wl['desc']}, m3det_{SENSITIVITY} detector cfg, {B} frames/step resident in HBM, "
                               f"one independent stream per GPU" + (" -- INPUT FROM PINNED HOST MEMORY (PCIe-inclusive)" if a.host_input else ""),
                   "frame": [W, H], "window": n, "frames_per_step": B,
                   "parallelism": f"streams x{world}"},
        "detections": {"raw_segments": tot[0], "nms_segments": tot[1], "lit_pixels": tot[2]},
    }
    if rank == 0:
        if prof:
            ms, launches = prof["stack_threshold"]
            per_launch_frames = B  # one launch covers the whole pass (max_batch == batch)
            alg_bytes = (n + 1) * H * W * per_launch_frames
            avg_s = ms / max(launches, 1) * 1e-3
            achieved = alg_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
            traffic = None
            tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{a.workload}.json")
            if os.path.exists(tpath):  # rocprofv3 PMC passes of this workload (tools/profile_round.sh), scaled to this batch
                t = json.load(open(tpath))
                if t.get("stack_threshold_bytes_per_launch"):
                    traffic = int(t["stack_threshold_bytes_per_launch"] * B / t.get("frames_per_launch", B))
            out["roofline"] = {"bound": "hbm", "kernel": "stack_threshold", "achieved": achieved, "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "avg_launch_ms": ms / max(launches, 1), "launches": launches,
                               "launches_in_flight": 2,  # consecutive passes alternate between two main streams: the durations overlap

                               "algorithmic_bytes_per_launch": alg_bytes}
            out["kernels_ms_per_step"] = {k: v[0] / v[1] for k, v in prof.items() if v[1]}  # mean span per timed pass
        if world == 1 and a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(frames, n, mask, a.cpu_seconds)
        if world == 1 and a.parity_frames > 0:
            out["parity"] = parity_check(frames, n, mask, local, a.parity_frames)
        print(json.dumps(out), flush=True)
    (ctx.host_free if a.host_input else ctx.device_free)(pool)
    det.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

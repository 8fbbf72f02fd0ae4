#!/bin/bash
# Round profile of a bench.py workload on the GPU box:  bash tools/profile_round.sh <tag> [workload]   (e.g. r02 C2)
#   1. rocprofv3 --kernel-trace --stats          -> per-kernel durations (profiles/<tag>_rocprofv3_kernel_stats.csv)
#   2. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE    -> HBM traffic per launch, separate passes (MI355X_MICROARCH.md, HBM section)
# Outputs land in gpurun_out/profile_<tag>/; copy the summaries into profiles/.
set -u
TAG=${1:-r01}
WL=${2:-C2}
NAME=${NAME:-$WL}   # (file names: NAME=C2n25 EXTRA="--window 25" bash tools/profile_round.sh r06 C2)
OUT=gpurun_out/profile_${TAG}_$NAME
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --workload $WL --steps 24 --warmup 6 --passes-per-step 1 --no-secondary --cpu-seconds 0 --parity-frames 0 --standalone-passes 0 --no-pmc ${EXTRA:-}"
# (C5 flips to the dense output format after its first collected pass: pinned for the profile so that every launch is the same kernel)
if [ "$WL" = "C5" ]; then export MDH_DENSE=1; fi
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o r --output-format csv -- python $ARGS > "$OUT/stats.log" 2>&1
# the same passes one at a time (nothing overlaps): the average duration of the window kernel here is what roofline.launch_ms / frac use
rocprofv3 --kernel-trace --stats -d "$OUT/stats_d1" -o r --output-format csv -- python $ARGS --depth 1 > "$OUT/stats_d1.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o r --output-format csv -- python $ARGS > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o r --output-format csv -- python $ARGS > "$OUT/write.log" 2>&1
python - "$OUT" "$TAG" "$WL" "$NAME" <<'PY'
import csv, glob, json, sys, collections, shutil, os
sys.path.insert(0, os.getcwd())
out, tag, wl, name = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None
stats = one(f"{out}/stats/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, f"{out}/{tag}_rocprofv3_kernel_stats_{name}.csv")
stats1 = one(f"{out}/stats_d1/**/*kernel_stats.csv")
if stats1:
    shutil.copy(stats1, f"{out}/{tag}_rocprofv3_kernel_stats_{name}_depth1.csv")
def per_kernel(path, counter):
    acc, cnt, seen = collections.defaultdict(float), collections.Counter(), set()
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        k = row["Kernel_Name"].split("(")[0]
        acc[k] += float(row["Counter_Value"])
        if (k, row["Dispatch_Id"]) not in seen:
            seen.add((k, row["Dispatch_Id"])); cnt[k] += 1
    return {k: (acc[k] / cnt[k], cnt[k]) for k in acc}
fetch, write = one(f"{out}/fetch/**/*counter_collection.csv"), one(f"{out}/write/**/*counter_collection.csv")
import bench
res = {"note": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --workload {wl} at its default batch. FETCH_SIZE (KB) is doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-byte requests of wide coalesced "
               "reads at 64 bytes); WRITE_SIZE taken as reported (uncalibrated).", "kernels": {}}
if fetch and write:
    F, Wr = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    for k in F:
        if k.startswith("__amd") or k not in Wr:
            continue
        res["kernels"][k] = {"FETCH_SIZE_KB_raw": round(F[k][0], 1), "WRITE_SIZE_KB_raw": round(Wr[k][0], 1), "launches": F[k][1],
                             "hbm_read_bytes_per_launch": int(F[k][0] * 1024 * 2), "hbm_write_bytes_per_launch": int(Wr[k][0] * 1024)}
        if "stack_threshold" in k:
            res["stack_threshold_bytes_per_launch"] = res["kernels"][k]["hbm_read_bytes_per_launch"] + res["kernels"][k]["hbm_write_bytes_per_launch"]
res["frames_per_launch"] = bench.WORKLOADS[wl]["batch"]
json.dump(res, open(f"{out}/pmc_traffic_{name}.json", "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "note"})[:1500])
PY
head -14 "$OUT/${TAG}_rocprofv3_kernel_stats_${NAME}.csv" 2>/dev/null | cut -c1-160

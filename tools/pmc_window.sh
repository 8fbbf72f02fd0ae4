#!/bin/bash
# SQ counters of the window kernel (stand-alone passes, nothing overlapping).  Usage on the GPU box: bash tools/pmc_window.sh [outdir]
# Counter passes are separate rocprofv3 runs with --pmc only (no tracing flags besides --kernel-trace).
set -u
OUT=${1:-gpurun_out/pmc_window}
mkdir -p "$OUT"
export TMPDIR=/tmp
run() {  # name, counters...
    local name=$1; shift
    rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o r --output-format csv -- python tools/standalone_kernel_times.py > "$OUT/$name.log" 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("sq1", "sq2", "grbm"):
    files = glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (k, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key); cnt[k] += 1
    print("==", name)
    for k in acc:
        if "stack_threshold" in k or "morph_close" in k or "hough" in k:
            print(k, "dispatches", cnt[k], {c: round(v / max(cnt[k], 1)) for c, v in acc[k].items()})
PY

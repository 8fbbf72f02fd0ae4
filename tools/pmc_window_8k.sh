#!/bin/bash
# SQ / write counters of the window kernel on 8K frames at m3det_high against m3det_normal thresholds (one pass in flight).
# On the GPU box:  bash tools/pmc_window_8k.sh [outdir]      -> profiles/r02_pmc_window_8k_high_vs_normal.txt
# Counter passes are separate rocprofv3 runs with --pmc only (no tracing flags besides --kernel-trace).
set -u
OUT=${1:-gpurun_out/pmc_window_8k}; mkdir -p "$OUT"
export TMPDIR=/tmp
for sens in high normal; do
ARGS="bench.py --workload C5 --unique 50 --batch 128 --sensitivity $sens --steps 4 --warmup 2 --cpu-seconds 0 --parity-frames 0 --standalone-passes 0 --depth 1"
run() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/${sens}_$name" -o r --output-format csv -- python $ARGS > "$OUT/${sens}_$name.log" 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES
run wr WRITE_SIZE
run rd FETCH_SIZE
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sens in ("high", "normal"):
    for name in ("sq1", "sq2", "wr", "rd"):
        files = glob.glob(f"{out}/{sens}_{name}/**/*counter_collection.csv", recursive=True)
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
        for f in files:
            for row in csv.DictReader(open(f)):
                k = row["Kernel_Name"].split("(")[0][:40]
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                if (k, row["Dispatch_Id"]) not in seen:
                    seen.add((k, row["Dispatch_Id"])); cnt[k] += 1
        for k in acc:
            if "stack_threshold" in k:
                print(sens, name, "dispatches", cnt[k], {c: round(v / max(cnt[k], 1)) for c, v in acc[k].items()})
PY

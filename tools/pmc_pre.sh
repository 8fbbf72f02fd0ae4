#!/bin/bash
# SQ counters of the loader-preprocessing kernel (k_preprocess_rows: 3840x2160x3 BGR -> 960x540 gray, 64 output frames per launch,
# everything resident in HBM: tools/bench_pre_batch.py).  Separate rocprofv3 --pmc passes (no tracing flags besides --kernel-trace).
#   bash tools/pmc_pre.sh [outdir]
set -u
OUT=${1:-gpurun_out/pmc_pre}; mkdir -p "$OUT"; export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o r --output-format csv -- python tools/bench_pre_batch.py > "$OUT/$name.log" 2>&1; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES
run c SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("a", "b", "c"):
    acc = collections.defaultdict(float); seen = set(); dur = []
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_preprocess_rows" not in row["Kernel_Name"]: continue
            if int(row.get("Grid_Size_Z", row.get("Grid_Size", "0")) or 0) not in (64,): pass
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); seen.add(row["Dispatch_Id"])
    for f in glob.glob(f"{out}/{name}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_preprocess_rows" in row["Kernel_Name"]: dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    n = max(len(seen), 1); wc = acc.get("SQ_WAVE_CYCLES", 1) / n
    print(f"pass {name}: {n} launches of k_preprocess_rows, mean duration under the counters {sum(dur) / max(len(dur), 1) / 1e3:.1f} us")
    for c, v in sorted(acc.items()): print(f"   {c:28s} {v / n:16.0f}   {v / n / wc:6.3f} of SQ_WAVE_CYCLES")
PY

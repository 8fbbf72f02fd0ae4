#!/bin/bash
# SQ counters of the window kernel that show the CU's vector-memory front end (round 4): parked / issue-stalled / active wave cycles and
# the texture addresser's FIFO-full cycles.  Separate rocprofv3 --pmc passes, stand-alone passes of tools/standalone_kernel_times.py.
#   bash tools/pmc_window_frontend.sh [outdir]      (MDH_LIB=... selects another build, e.g. one made with -DWIDE_AUX=0)
set -u
OUT=${1:-gpurun_out/pmc_frontend}; mkdir -p "$OUT"; export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o r --output-format csv -- python tools/standalone_kernel_times.py > "$OUT/$name.log" 2>&1; }
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run b SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("a", "b"):
    acc = collections.defaultdict(float); seen = set(); dur = []
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "stack_threshold" not in row["Kernel_Name"]: continue
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); seen.add(row["Dispatch_Id"])
    for f in glob.glob(f"{out}/{name}/**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "stack_threshold" in row["Kernel_Name"]: dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    n = max(len(seen), 1); wc = acc.get("SQ_WAVE_CYCLES", 1) / n
    print(f"pass {name}: {n} launches of k_stack_threshold_wide (1080p, n = " + __import__('os').environ.get('MDH_PROBE_N', '8') + f", 1024 frames), mean duration under the counters {sum(dur) / max(len(dur), 1) / 1e3:.1f} us")
    for c, v in sorted(acc.items()): print(f"   {c:28s} {v / n:16.0f}   {v / n / wc:6.3f} of SQ_WAVE_CYCLES")
PY

#!/bin/bash
# SQ / cache counters of the morphology kernel on the 8K m3det_high stream (dense binaries).  bash tools/pmc_morph.sh [outdir]
set -u
OUT=${1:-gpurun_out/pmc_morph}
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="bench.py --workload C5 --unique 50 --steps 6 --warmup 2 --cpu-seconds 0 --parity-frames 0 --standalone-passes 0 --depth 1"
run() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/$name" -o r --output-format csv -- python $ARGS > "$OUT/$name.log" 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
run tc1 TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
run fetch FETCH_SIZE
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for name in ("sq1", "sq2", "tc1", "fetch"):
    files = glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
    for f in files:
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:60]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            key = (k, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key); cnt[k] += 1
    print("==", name)
    for k in acc:
        if "morph" in k or "stack_threshold" in k:
            print(k, "dispatches", cnt[k], {c: round(v / max(cnt[k], 1)) for c, v in acc[k].items()})
PY

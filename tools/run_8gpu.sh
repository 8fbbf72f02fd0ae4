#!/bin/bash
# One node, N GPUs (default 8), one process per GPU over RCCL/xGMI: bench.py in its three sharding modes, each launched the way the
# driver launches `--gpus N` (torch.distributed.run, rendezvous on 127.0.0.1).  NOT run on the build's one-GPU boxes: no scaling
# number in this repository comes from here until a node with N GPUs has run it.
#
#   bash tools/run_8gpu.sh [--dry-run] [N] [out_dir]
#
# --dry-run: the SAME three commands with N ranks on however many devices the box has (one is enough): the ranks share devices
# round-robin, no RCCL communicator can form (RCCL refuses two ranks per device), so barrier / max / gathers go through the
# rendezvous store and the band path exchanges host buffers.  Small passes; proves the structural path the driver launches at
# N = 8 end to end (launcher -> rendezvous -> every mode's sharding -> gather -> one JSON line with parity), not the scaling.
#
# Every rank prints, before its first pass: the HIP device it sits on, what the RCCL communicator of the band path says about
# itself (ranks, rank, device), and the message sizes of its collectives.  It fails loudly -- non-zero exit, text on stderr --
# when fewer GPUs are visible than ranks, when the communicator does not form within MDH_RCCL_TIMEOUT seconds, when it holds
# another number of ranks than WORLD_SIZE (ncclCommCount), or when a pass that waits on a peer does not complete in that time.
set -u
DRY=0
if [ "${1:-}" = "--dry-run" ]; then DRY=1; shift; fi
N=${1:-8}
OUT=${2:-gpurun_out/run_${N}gpu}
cd "$(dirname "$0")/.."
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-12} MDH_RCCL_TIMEOUT=${MDH_RCCL_TIMEOUT:-120} MDH_BENCH_VERBOSE_RANKS=1
have=$(python - <<'PY'
import ctypes, sys
try:
    hip = ctypes.CDLL("libamdhip64.so")
    n = ctypes.c_int(0)
    print(n.value if hip.hipGetDeviceCount(ctypes.byref(n)) else n.value)
except OSError:
    print(0)
PY
)
if [ "$DRY" = 1 ]; then
    [ "$have" -ge 1 ] || { echo "run_8gpu.sh --dry-run: no HIP device visible" >&2; exit 2; }
    export MDH_DIST_BACKEND=store
elif [ "$have" -lt "$N" ]; then
    echo "run_8gpu.sh: $N ranks asked for, $have HIP device(s) visible -- RCCL refuses two ranks on one device; nothing was run" >&2
    exit 2
fi
launch() {  # name, port, bench.py arguments...
    local name=$1 port=$2; shift 2
    echo "== $name: python -m torch.distributed.run --nproc-per-node $N bench.py --gpus $N $*"
    timeout 1800 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port "$port" \
        bench.py --gpus "$N" "$@" > "$OUT/$name.out" 2> "$OUT/$name.err"
    local rc=$?
    grep '^{' "$OUT/$name.out" > "$OUT/$name.json"   # (the launcher's and gloo's chatter stays in .out / .err)
    grep -h '^\[rank' "$OUT/$name.err" | sort | head -$((4 * N))
    if [ $rc -ne 0 ] || [ ! -s "$OUT/$name.json" ]; then
        echo "run_8gpu.sh: $name FAILED (rc $rc)" >&2
        tail -20 "$OUT/$name.err" >&2
        return 1
    fi
    python - "$OUT/$name.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"   {d['value']:.0f} {d['unit']} on {d['n_gpus']} GPUs, {d['ms_per_step']:.3f} ms/step, scaling {d['scaling']}, parity {d.get('parity', {}).get('verdict')}")
PY
}
fail=0
if [ "$DRY" = 1 ]; then
    launch streams 29611 --steps 4 --warmup 1 --batch 128 --passes-per-step 2 --unique 64 || fail=1
    launch bands_C5 29612 --mode bands --workload C5 --unique 16 --batch 16 --steps 3 --warmup 1 --cpu-seconds 0 --parity-frames 8 || fail=1
    launch timeshard_C2 29613 --mode timeshard --batch 64 --steps 16 --unique 64 --cpu-seconds 0 || fail=1
    exit $fail
fi
# BASELINE config 4: N independent 1080p streams, no data-path collective (the driver's scaling curve runs exactly this)
launch streams 29611 --steps 20 --warmup 5 || fail=1
# BASELINE config 5: ONE 7680x4320 stream tiled into N row bands: all-reduce of the noise sums + all-to-all of the bit planes (RCCL on the library's streams)
launch bands_C5 29612 --mode bands --workload C5 --unique 50 --batch 256 --steps 12 --warmup 2 --cpu-seconds 0 || fail=1
# north_star's sharding: ONE 1080p stream cut into N time windows: all-gather of the threshold table, gather of the results
launch timeshard_C2 29613 --mode timeshard --steps 16 --cpu-seconds 0 || fail=1
exit $fail

O=gpurun_out/r04_34; mkdir -p $O
A="--cpu-seconds 0 --no-pmc --no-secondary"
for rep in 1 2; do for sb in 1024 4096 8192; do MDH_TS_SCAN_BATCH=$sb python bench.py --mode timeshard $A > $O/ts_${sb}_$rep.json 2>> $O/err; python -c "
import json
d=json.loads(open('$O/ts_${sb}_$rep.json').read().strip().splitlines()[-1]);print('ts SB $sb',int(d['value']),round(d['ms_per_step'],3), (d.get('parity') or {}).get('verdict'), d.get('timeshard',{}).get('first_sweep_s'))"; done; done
timeout 600 python -m pytest tests/test_gpu_bench_contract.py -m gpu -q -k "time_window" > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -2

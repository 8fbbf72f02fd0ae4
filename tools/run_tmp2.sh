O=gpurun_out/r04_25; mkdir -p $O
A="--cpu-seconds 0 --no-pmc --no-secondary --parity-frames 0"
for rep in 1 2; do for th in 4 8 2; do MDH_TS_SCAN_THREADS=$th python bench.py --mode timeshard $A > $O/ts_${th}_$rep.json 2>> $O/err; python -c "
import json
d=json.loads(open('$O/ts_${th}_$rep.json').read().strip().splitlines()[-1]);print('ts threads $th',int(d['value']),round(d['ms_per_step'],3), d.get('timeshard',{}).get('first_sweep_s'))"; done; done

O=gpurun_out/r04_30; mkdir -p $O; export TMPDIR=/tmp
A="--cpu-seconds 0 --no-pmc --no-secondary"
timeout 900 python -m pytest tests/test_bands.py tests/test_gpu_baseline_configs.py -m gpu -q -x > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -3
for rep in 1 2; do for v in "new:X=1" "nocount:MDH_BANDS_COUNT_IN_EXPORT=0" "new_mainhi:MDH_PRIO_MAIN=1" "r3:MDH_BANDS_COUNT_IN_EXPORT=0 MDH_BANDS_SHARE_STREAMS=0"; do name=${v%%:*}; e=${v#*:}
env $e python bench.py --mode bands --workload C5 --unique 50 --batch 500 --steps 16 --warmup 3 $A > $O/bands_${name}_$rep.json 2>> $O/err
python -c "
import json
d=json.loads(open('$O/bands_${name}_$rep.json').read().strip().splitlines()[-1]);print('bands $name',int(d['value']),round(d['ms_per_step'],3), (d.get('parity') or {}).get('verdict'))"
done; done

O=gpurun_out/r04_19; mkdir -p $O; export TMPDIR=/tmp
A="--cpu-seconds 0 --no-pmc --no-secondary"
timeout 900 python -m pytest tests/test_timeshard.py tests/test_gpu_parity.py tests/test_gpu_per_frame_buffers.py -m gpu -q -x > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -4
for rep in 1 2; do
for v in "def=metdetpy_amd/libmetdet_hip.so" "nf256=metdetpy_amd/libmetdet_hip_nf256.so"; do name=${v%%=*}; lib=${v#*=}
MDH_LIB=$PWD/$lib python bench.py --mode timeshard $A > $O/ts_${name}_$rep.json 2>> $O/err
MDH_LIB=$PWD/$lib python bench.py --mode bands --workload C5 --unique 50 --batch 500 --steps 12 --warmup 3 $A > $O/bands_${name}_$rep.json 2>> $O/err
python -c "
import json
for k in ('ts','bands'):
    d=json.loads(open('$O/%s_${name}_$rep.json'%k).read().strip().splitlines()[-1]);print(k,'$name',int(d['value']),round(d['ms_per_step'],3),(d.get('parity') or {}).get('verdict'), d.get('timeshard',{}).get('first_sweep_s'))"
done; done
bash tools/run_tmp.sh r04_19 "def=metdetpy_amd/libmetdet_hip.so nf256=metdetpy_amd/libmetdet_hip_nf256.so" "C1s C2"

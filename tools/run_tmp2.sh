O=gpurun_out/r04_23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_gpu_per_frame_buffers.py tests/test_timeshard.py -m gpu -q -x > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -3
bash tools/run_tmp.sh r04_23 "dy4=metdetpy_amd/libmetdet_hip.so" "C3 C1s C2"
python bench.py --protocol per-frame --steps 8000 --no-secondary --cpu-seconds 0 --parity-frames 64 > $O/pf.json 2>>$O/err; python -c "
import json;d=json.loads(open('$O/pf.json').read().strip().splitlines()[-1]);print('pf',int(d['value']),d['per_frame']['us_per_frame'],d['parity']['verdict'])"

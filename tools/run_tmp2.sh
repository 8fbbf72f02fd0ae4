O=gpurun_out/r04_12; mkdir -p $O
for k in pinned recycled pageable pinned; do python bench.py --protocol per-frame --frames $k --steps 8000 --no-secondary --cpu-seconds 0 --parity-frames 0 > $O/pf_$k.json 2>>$O/err; python -c "
import json;d=json.loads(open('$O/pf_$k.json').read().strip().splitlines()[-1]);print('$k',int(d['value']),d['per_frame']['us_per_frame'],d['per_frame']['update_stats'])"; done
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -30

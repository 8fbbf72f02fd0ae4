O=gpurun_out/r04_16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hough_planes.py tests/test_gpu_hough_accumulator.py tests/test_gpu_baseline_configs.py -m gpu -q -x > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -4
C="--no-pmc --cpu-seconds 0 --parity-frames 0 --no-profile --no-secondary --steps 6 --warmup 1"
for rep in 1 2; do for wl in C1s C2 C3; do for v in "tiny40:" "tiny0:MDH_HOUGH_LDS_TINY=0" "tiny32:MDH_HOUGH_LDS_TINY=32768" "tiny26:MDH_HOUGH_LDS_TINY=26624" "tiny40s64:MDH_HOUGH_LDS_SMALL=65536"; do
   name=${v%%:*}; e=${v#*:}
   env $e python bench.py --workload $wl $C > $O/x_${wl}_${name}_$rep.json 2>>$O/err
   python -c "
import json;d=json.loads(open('$O/x_${wl}_${name}_$rep.json').read().strip().splitlines()[-1]);print('$wl $name $rep', int(d['value']), round(d['ms_per_step']/d['config']['passes_per_step'],3))"
done; done; done

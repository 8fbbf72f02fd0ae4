O=gpurun_out/r04_36; mkdir -p $O
C="--no-pmc --cpu-seconds 0 --parity-frames 0 --no-secondary --gpus 1 --steps 20 --warmup 5"
for rep in 1 2 3 4 5; do python bench.py $C > $O/d_$rep.json 2>>$O/err; python -c "
import json;d=json.loads(open('$O/d_$rep.json').read().strip().splitlines()[-1]);r=d['roofline'];print('run $rep', int(d['value']), round(r['launch_ms'],3), round(r['frac'],3), round(r['frac_at_pipeline_pace'],3))"; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -4; done

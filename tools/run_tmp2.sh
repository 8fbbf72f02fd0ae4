O=gpurun_out/r04_21; mkdir -p $O
bash tools/run_tmp.sh r04_21 "def=metdetpy_amd/libmetdet_hip.so u2=metdetpy_amd/libmetdet_hip_u2.so" "C3"
C="--no-pmc --cpu-seconds 0 --parity-frames 0 --standalone-passes 8 --no-secondary --steps 6 --warmup 1"
for rep in 1 2; do for v in "nb48:" "nb16:MDH_NOISE_BLOCKS=16" "nb8:MDH_NOISE_BLOCKS=8"; do
   name=${v%%:*}; e=${v#*:}
   env $e python bench.py --workload C1s $C > $O/x_C1s_${name}_$rep.json 2>>$O/err
   python -c "
import json;d=json.loads(open('$O/x_C1s_${name}_$rep.json').read().strip().splitlines()[-1]);print('C1s $name $rep', int(d['value']), round(d['ms_per_step']/d['config']['passes_per_step'],3), round(d['kernels_ms_standalone']['noise_sums'],3))"
done; done

O=gpurun_out/r04_35; mkdir -p $O
C="--no-pmc --cpu-seconds 0 --parity-frames 0 --no-profile --no-secondary --steps 6 --warmup 1"
for rep in 1 2 3; do for v in "hp3=metdetpy_amd/libmetdet_hip.so" "hp0=metdetpy_amd/libmetdet_hip_hp0.so" "hp1=metdetpy_amd/libmetdet_hip_hp1.so"; do name=${v%%=*}; lib=${v#*=}
for wl in C1s C3; do MDH_LIB=$PWD/$lib python bench.py --workload $wl $C > $O/${wl}_${name}_$rep.json 2>>$O/err; python -c "
import json;d=json.loads(open('$O/${wl}_${name}_$rep.json').read().strip().splitlines()[-1]);print('$wl $name $rep', int(d['value']))"; done; done; done

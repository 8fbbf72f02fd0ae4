O=gpurun_out/r04_31; mkdir -p $O; export TMPDIR=/tmp
A="--cpu-seconds 0 --no-pmc --no-secondary --parity-frames 0"
python bench.py --mode bands --workload C5 --unique 50 --batch 500 --steps 8 --warmup 3 $A > /dev/null 2>&1
for rep in 1 2; do for v in "new:X=1" "priv:MDH_MAIN_PRIVATE_QUEUE=1"; do name=${v%%:*}; e=${v#*:}
env $e python bench.py --mode bands --workload C5 --unique 50 --batch 500 --steps 16 --warmup 3 $A > $O/bands_${name}_$rep.json 2>> $O/err
python -c "
import json
d=json.loads(open('$O/bands_${name}_$rep.json').read().strip().splitlines()[-1]);print('bands $name',int(d['value']),round(d['ms_per_step'],3))"
env $e python bench.py --workload C2 --steps 6 --warmup 1 --no-profile $A > $O/c2_${name}_$rep.json 2>> $O/err
python -c "
import json
d=json.loads(open('$O/c2_${name}_$rep.json').read().strip().splitlines()[-1]);print('C2 $name',int(d['value']))"
env $e python bench.py --workload C1s --steps 6 --warmup 1 --no-profile $A > $O/c1s_${name}_$rep.json 2>> $O/err
python -c "
import json
d=json.loads(open('$O/c1s_${name}_$rep.json').read().strip().splitlines()[-1]);print('C1s $name',int(d['value']))"
done; done

# experiment: pipeline depth (MDH_SLOTS) x Hough sets (MDH_HOUGH_SETS)
run() { timeout 200 python bench.py --steps 40 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(round(d['value']),round(d['ms_per_step'],4),{k:round(v,3) for k,v in d['kernels_ms_per_step'].items()},d['parity']['verdict'])"; }
for cfg in "4 2" "4 2" "6 2" "4 1"; do set -- $cfg
echo "slots=$1 sets=$2"; MDH_SLOTS=$1 MDH_HOUGH_SETS=$2 run; done

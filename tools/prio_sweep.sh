# experiment: Hough set count x stream priorities (main noise hough: -1 low, 0 normal, 1 high; unset = plain streams)
run() { timeout 200 python bench.py --steps 30 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(round(d['value']),round(d['ms_per_step'],4),{k:round(v,3) for k,v in d['kernels_ms_per_step'].items()},d['parity']['verdict'])"; }
echo "plain sets=1"; run
echo "plain sets=1 again"; run
echo "plain sets=2"; MDH_HOUGH_SETS=2 run

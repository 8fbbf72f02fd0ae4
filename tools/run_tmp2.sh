O=gpurun_out/r04_20; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_bands.py -m gpu -q -x > $O/pytest.log 2>&1; grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/pytest.log | tail -4
C="--no-pmc --cpu-seconds 0 --parity-frames 0 --standalone-passes 12 --no-secondary --steps 5 --warmup 1"
for rep in 1 2; do for wl in C3 C5 C2; do for v in "wpfauto:" "wpf1:MDH_MORPH_WPF=1"; do
   name=${v%%:*}; e=${v#*:}
   X=""; if [ $wl = C5 ]; then X="--unique 50"; fi
   env $e python bench.py --workload $wl $X $C > $O/${wl}_${name}_$rep.json 2>>$O/err
done; done; done
python - $O <<'P'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+'/C*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f,'ERR'); continue
    r=d['roofline']; print(f.split('/')[-1], int(d['value']), round(r.get('launch_ms',0),3), round(r['frac'],3), round(r['frac_at_pipeline_pace'],3), {k:round(v,3) for k,v in d.get('kernels_ms_standalone',{}).items()})
P
A="bench.py --workload C5 --unique 50 --steps 16 --warmup 6 --passes-per-step 1 --no-secondary --cpu-seconds 0 --parity-frames 0 --standalone-passes 0 --no-pmc"
rocprofv3 --kernel-trace --stats -d $O/d1_C5 -o r --output-format csv -- python $A --depth 1 > $O/d1_C5.log 2>&1
python - $(find $O/d1_C5 -name "*kernel_stats.csv" | head -1) <<'P'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(r['Name'][:58].ljust(58), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1000:9.1f} us", r['Percentage'])
P

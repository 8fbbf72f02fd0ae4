cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log
python tools/bench_per_frame.py > $O/per_frame.json 2> $O/err
python bench.py --protocol per-frame --cpu-seconds 0 > $O/bench_pf.json 2>> $O/err
PF_N=400 rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace -o pf --output-format csv -- python tools/bench_per_frame.py > $O/trace.log 2>&1
python - <<'PY' > $O/chain.txt 2>&1
import csv, glob
k = glob.glob('gpurun_out/r02q/trace/**/*kernel_trace.csv', recursive=True)
ev = []
for r in csv.DictReader(open(k[0])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40]))
ev.sort()
mid = len(ev) // 2
t0 = ev[mid][0]
for s, e, nme in ev[mid:mid + 80]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {nme}")
PY
B="--cpu-seconds 0 --parity-frames 64 --standalone-passes 6 --steps 300"
python bench.py --workload C3 --batch 1024 $B > $O/C3_1024.json 2>> $O/err
MDH_MAIN_STREAMS=1 python bench.py --workload C3 --batch 1024 $B > $O/C3_1024_m1.json 2>> $O/err
MDH_POST_CUS=2 MDH_POST_MASKED=1 python bench.py --workload C3 --batch 1024 $B > $O/C3_1024_cus2.json 2>> $O/err
MDH_POST_CUS=4 MDH_POST_MASKED=1 python bench.py --workload C3 --batch 1024 $B > $O/C3_1024_cus4.json 2>> $O/err
MDH_SLOTS=8 python bench.py --workload C3 --batch 1024 $B > $O/C3_1024_s8.json 2>> $O/err
MDH_SKIP_HOUGH=1 python bench.py --workload C3 --batch 1024 --cpu-seconds 0 --parity-frames 0 --standalone-passes 6 --steps 300 > $O/C3_1024_nohough.json 2>> $O/err
python bench.py --workload C5 --unique 50 $B > $O/C5.json 2>> $O/err
python bench.py --cpu-seconds 0 --parity-frames 64 --standalone-passes 6 --steps 1000 > $O/C2.json 2>> $O/err
tail -3 $O/pytest.log

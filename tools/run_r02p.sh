cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pinned" 2>&1 | tail -15 > $O/pytest_pinned.log
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log
python tools/bench_per_frame.py > $O/per_frame.json 2> $O/err
MDH_LEAN_POLL=0 python tools/bench_per_frame.py > $O/per_frame_nopoll.json 2>> $O/err
MDH_LEAN_DIRECT=0 python tools/bench_per_frame.py > $O/per_frame_nodirect.json 2>> $O/err
python bench.py --protocol per-frame --cpu-seconds 0 > $O/bench_pf.json 2>> $O/err
PF_N=400 rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace -o pf --output-format csv -- python tools/bench_per_frame.py > $O/trace.log 2>&1
python - <<'PY' > $O/chain.txt 2>&1
import csv, glob
k = glob.glob('gpurun_out/r02p/trace/**/*kernel_trace.csv', recursive=True)
m = glob.glob('gpurun_out/r02p/trace/**/*memory_copy_trace.csv', recursive=True)
ev = []
for r in csv.DictReader(open(k[0])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40]))
if m:
    for r in csv.DictReader(open(m[0])):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '') ))
ev.sort()
mid = len(ev) // 2
t0 = ev[mid][0]
for s, e, nme in ev[mid:mid + 100]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {nme}")
PY
B="--cpu-seconds 0 --parity-frames 64 --standalone-passes 6 --steps 300"
python bench.py --workload C3 --batch 1024 $B > $O/C3_1024.json 2>> $O/err
MDH_HOUGH_SPLIT=0 python bench.py --workload C3 --batch 1024 $B > $O/C3_1024_nosplit.json 2>> $O/err
python bench.py --workload C5 --unique 50 $B > $O/C5.json 2>> $O/err
MDH_HOUGH_SPLIT=0 python bench.py --workload C5 --unique 50 $B > $O/C5_nosplit.json 2>> $O/err
python bench.py --cpu-seconds 0 --parity-frames 64 --standalone-passes 6 --steps 1000 > $O/C2.json 2>> $O/err
tail -3 $O/pytest_pinned.log $O/pytest.log

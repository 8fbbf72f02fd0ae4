cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log
python tools/bench_per_frame.py > $O/per_frame.json 2> $O/err
PF_N=400 rocprofv3 --kernel-trace --memory-copy-trace -d $O/trace -o pf --output-format csv -- python tools/bench_per_frame.py > $O/trace.log 2>&1
python - <<'PY' > $O/chain.txt 2>&1
import csv, glob
k = glob.glob('gpurun_out/r02o/trace/**/*kernel_trace.csv', recursive=True)
m = glob.glob('gpurun_out/r02o/trace/**/*memory_copy_trace.csv', recursive=True)
ev = []
for r in csv.DictReader(open(k[0])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:40]))
for r in csv.DictReader(open(m[0])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '') ))
ev.sort()
# print a window of events in the middle
mid = len(ev) // 2
t0 = ev[mid][0]
for s, e, nme in ev[mid:mid + 120]:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f}  {nme}")
PY
tail -3 $O/pytest.log

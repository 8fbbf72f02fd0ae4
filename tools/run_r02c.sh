cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.log
python bench.py --cpu-seconds 0 > $O/bench_C2.json 2> $O/bench_C2.err
python bench.py --workload C3 --cpu-seconds 0 > $O/bench_C3.json 2> $O/bench_C3.err
python bench.py --workload C3 --batch 1024 --cpu-seconds 0 --parity-frames 0 > $O/bench_C3_1024.json 2> $O/bench_C3_1024.err
python bench.py --workload C5 --cpu-seconds 0 --unique 50 > $O/bench_C5.json 2> $O/bench_C5.err
python bench.py --workload C1s --cpu-seconds 0 > $O/bench_C1s.json 2> $O/bench_C1s.err
WL=C2 python tools/hough_frame_times.py > $O/hough_C2.txt 2>&1
WL=C3 python tools/hough_frame_times.py > $O/hough_C3.txt 2>&1
WL=C5 python tools/hough_frame_times.py > $O/hough_C5.txt 2>&1
rocm-smi --showmeminfo vram > $O/smi.txt 2>&1
tail -3 $O/pytest.log

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02a/pytest.log
python bench.py > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
python bench.py --workload C3 --cpu-seconds 6 > gpurun_out/r02a/bench_C3.json 2> gpurun_out/r02a/bench_C3.err
WL=C2 python tools/hough_frame_times.py > gpurun_out/r02a/hough_C2.txt 2>&1
WL=C3 python tools/hough_frame_times.py > gpurun_out/r02a/hough_C3.txt 2>&1
tail -5 gpurun_out/r02a/pytest.log

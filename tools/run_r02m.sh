cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hough_planes.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py tests/test_bands.py tests/test_timeshard.py -m gpu -x -q 2>&1 | tail -30 > $O/pytest.log
B="--cpu-seconds 0 --parity-frames 64 --standalone-passes 6 --steps 300"
python bench.py --workload C3 --batch 1024 $B > $O/C3_1024.json 2> $O/err
MDH_HOUGH_SETS=2 python bench.py --workload C3 --batch 1024 $B > $O/C3_1024_sets2.json 2>> $O/err
python bench.py --workload C5 --unique 50 $B > $O/C5.json 2>> $O/err
MDH_HOUGH_SETS=2 python bench.py --workload C5 --unique 50 $B > $O/C5_sets2.json 2>> $O/err
WL=C3 python tools/hough_frame_times.py > $O/hough_C3.txt 2>&1
WL=C5 python tools/hough_frame_times.py > $O/hough_C5.txt 2>&1
python bench.py --mode timeshard --cpu-seconds 0 > $O/timeshard.json 2>> $O/err
tail -3 $O/pytest.log

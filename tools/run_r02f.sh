cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.log
B="--cpu-seconds 0 --parity-frames 0 --steps 400 --standalone-passes 8"
MDH_HOST_TRACE=1 python bench.py --protocol per-frame --steps 5000 --cpu-seconds 0 --parity-frames 64 > $O/perframe.json 2> $O/perframe.err
python bench.py $B > $O/C2.json 2> $O/C2.err
python bench.py --workload C3 --batch 1024 $B > $O/C3_1024.json 2> $O/C3_1024.err
WL=C2 python tools/hough_frame_times.py > $O/hough_C2.txt 2>&1
WL=C3 python tools/hough_frame_times.py > $O/hough_C3.txt 2>&1
tail -3 $O/pytest.log

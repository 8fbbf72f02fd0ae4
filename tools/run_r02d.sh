cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hough_planes.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -30 > $O/pytest.log
B="--cpu-seconds 0 --parity-frames 0 --steps 400 --standalone-passes 8"
python bench.py $B > $O/C2.json 2> $O/C2.err
MDH_MAIN_STREAMS=1 python bench.py $B > $O/C2_main1.json 2> $O/C2_main1.err
MDH_SKIP_HOUGH=1 python bench.py $B > $O/C2_nohough.json 2> $O/C2_nohough.err
MDH_SLOTS=8 python bench.py $B > $O/C2_slots8.json 2> $O/C2_slots8.err
MDH_HOUGH_LDS_SMALL=54000 python bench.py $B > $O/C2_lds54.json 2> $O/C2_lds54.err
python bench.py --workload C3 $B > $O/C3.json 2> $O/C3.err
MDH_WIDE16_MAXN=16 python bench.py --workload C3 $B > $O/C3_w16.json 2> $O/C3_w16.err
MDH_WIDE16_MAXN=16 python bench.py --workload C3 --batch 1024 $B > $O/C3_w16_1024.json 2> $O/C3_w16_1024.err
WL=C2 python tools/hough_frame_times.py > $O/hough_C2.txt 2>&1
WL=C3 python tools/hough_frame_times.py > $O/hough_C3.txt 2>&1
tail -3 $O/pytest.log

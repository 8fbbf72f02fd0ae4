cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hough_planes.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -x -q 2>&1 | tail -30 > $O/pytest.log
B="--cpu-seconds 0 --parity-frames 0 --steps 400 --standalone-passes 8"
python bench.py $B > $O/C2.json 2> $O/C2.err
MDH_HOUGH_GRID_SMALL=512 python bench.py $B > $O/C2_g512.json 2> $O/C2_g512.err
MDH_HOUGH_GRID_SMALL=2048 MDH_HOUGH_GRID_LARGE=2048 python bench.py $B > $O/C2_g2048.json 2> $O/C2_g2048.err
MDH_LIB=$PWD/metdetpy_amd/libmetdet_hip_prio0.so python bench.py $B > $O/C2_prio0.json 2> $O/C2_prio0.err
MDH_HOUGH_SETS=1 python bench.py $B > $O/C2_sets1.json 2> $O/C2_sets1.err
MDH_SKIP_HOUGH=1 python bench.py $B > $O/C2_nohough.json 2> $O/C2_nohough.err
python bench.py --workload C3 --batch 1024 $B > $O/C3_1024.json 2> $O/C3_1024.err
python bench.py --protocol per-frame --steps 5000 --cpu-seconds 0 --parity-frames 64 > $O/perframe.json 2> $O/perframe.err
tail -3 $O/pytest.log

cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bench_contract.py tests/test_cv2_pinned.py tests/test_visu_golden.py tests/test_bands.py -m gpu -x -q 2>&1 | tail -40 > $O/pytest.log
python bench.py --mode bands --workload C5 --unique 50 --cpu-seconds 0 > $O/bands_C5.json 2> $O/bands_C5.err
python bench.py --mode timeshard --cpu-seconds 0 > $O/timeshard_C2.json 2> $O/timeshard_C2.err
python bench.py --workload C1e --cpu-seconds 0 > $O/C1e.json 2> $O/C1e.err
tail -5 $O/pytest.log

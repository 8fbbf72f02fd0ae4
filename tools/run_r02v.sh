cd $GRAFT_REPO_ROOT
O=gpurun_out/r02v; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hough_accumulator.py -m gpu -q 2>&1 | tail -40 > $O/pytest_acc.log
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log
python bench.py --mode timeshard --cpu-seconds 0 > $O/timeshard.json 2> $O/err
python bench.py --mode bands --workload C5 --unique 50 --cpu-seconds 0 > $O/bands.json 2>> $O/err
tail -8 $O/pytest_acc.log; tail -3 $O/pytest.log

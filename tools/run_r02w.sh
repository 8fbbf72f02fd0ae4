cd $GRAFT_REPO_ROOT
O=gpurun_out/r02w; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest.log
python bench.py --mode timeshard --cpu-seconds 0 > $O/timeshard.json 2> $O/err
python bench.py --hough-mode accumulator --cpu-seconds 0 --steps 600 > $O/C2_acc.json 2>> $O/err
python bench.py --workload C3 --hough-mode accumulator --cpu-seconds 0 --steps 200 > $O/C3_acc.json 2>> $O/err
python bench.py --workload C5 --unique 50 --hough-mode accumulator --cpu-seconds 0 --steps 200 > $O/C5_acc.json 2>> $O/err
tail -6 $O/pytest.log

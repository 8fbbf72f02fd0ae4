cd $GRAFT_REPO_ROOT
O=gpurun_out/r02x; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest.log
B="--cpu-seconds 0 --parity-frames 64 --standalone-passes 6"
python bench.py --workload C5 --unique 50 $B --steps 200 > $O/C5.json 2> $O/err
python bench.py --workload C3 $B --steps 200 > $O/C3.json 2>> $O/err
python bench.py $B --steps 1000 > $O/C2.json 2>> $O/err
python bench.py --mode bands --workload C5 --unique 50 --cpu-seconds 0 > $O/bands.json 2>> $O/err
python tools/bench_per_frame.py > $O/per_frame.json 2>> $O/err
tail -6 $O/pytest.log

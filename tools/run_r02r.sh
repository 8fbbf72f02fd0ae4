cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log
python bench.py --mode timeshard --cpu-seconds 0 > $O/timeshard.json 2> $O/err
python bench.py --mode timeshard --cpu-seconds 0 --host-input --steps 16 > $O/timeshard_host.json 2>> $O/err
B="--cpu-seconds 0 --parity-frames 64 --standalone-passes 6 --steps 200"
python bench.py --workload C3 --batch 2048 $B > $O/C3_2048.json 2>> $O/err
python bench.py --workload C3 --batch 1024 $B > $O/C3_1024.json 2>> $O/err
python bench.py --workload C5 --unique 50 $B > $O/C5.json 2>> $O/err
python tools/bench_per_frame.py > $O/per_frame.json 2>> $O/err
tail -3 $O/pytest.log

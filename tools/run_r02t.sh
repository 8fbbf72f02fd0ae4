cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
timeout 900 python -m pytest tests/test_timeshard.py tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | tail -15 > $O/pytest.log
python bench.py --mode timeshard --cpu-seconds 0 > $O/timeshard.json 2> $O/err
python bench.py --detector classic --cpu-seconds 0 --parity-frames 32 > $O/classic.json 2>> $O/err
MDH_CLASSIC_TB=0 python bench.py --detector classic --cpu-seconds 0 --parity-frames 32 --steps 300 > $O/classic_old.json 2>> $O/err
python bench.py --cpu-seconds 0 --parity-frames 64 --steps 1000 > $O/C2.json 2>> $O/err
tail -3 $O/pytest.log

cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s; mkdir -p $O
python tools/dbg_timeshard.py > $O/dbg.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "classic" 2>&1 | tail -15 > $O/pytest_classic.log
python bench.py --detector classic --cpu-seconds 0 --parity-frames 32 --steps 300 > $O/classic.json 2> $O/err
MDH_CLASSIC_TB=0 python bench.py --detector classic --cpu-seconds 0 --parity-frames 32 --steps 100 > $O/classic_old.json 2>> $O/err
tail -3 $O/pytest_classic.log

cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.log
python tools/bench_per_frame.py > $O/pf.json 2> $O/pf.err
python bench.py --protocol per-frame --steps 8000 --cpu-seconds 0 --parity-frames 128 > $O/perframe.json 2> $O/perframe.err
python bench.py > $O/bench_default.json 2> $O/bench_default.err
bash tools/profile_round.sh r02 C2 > $O/profile_C2.log 2>&1
tail -3 $O/pytest.log

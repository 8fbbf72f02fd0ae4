cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > $O/pytest.log
python tools/bench_per_frame.py > $O/pf.json 2> $O/pf.err
MDH_LEAN_NO_FUSE=1 python tools/bench_per_frame.py > $O/pf_nofuse.json 2> $O/pf_nofuse.err
python bench.py --protocol per-frame --steps 8000 --cpu-seconds 0 --parity-frames 128 > $O/perframe.json 2> $O/perframe.err
python -c "
import cProfile, pstats, sys
sys.argv=['bench.py','--mode','timeshard','--cpu-seconds','0','--steps','16','--parity-frames','0']
import bench
cProfile.run('bench.main()', '$O/ts.prof')
p=pstats.Stats('$O/ts.prof'); p.sort_stats('cumulative').print_stats(28)
" > $O/timeshard_profile.txt 2>&1
tail -3 $O/pytest.log

cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ac; mkdir -p $O
B="--cpu-seconds 0 --parity-frames 0 --standalone-passes 6"
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export MDH_LIB=$GRAFT_REPO_ROOT/metdetpy_amd/libmetdet_hip_old.so; else unset MDH_LIB; fi
  python bench.py --workload C3 $B --steps 300 > $O/C3_${v}_$rep.json 2>> $O/err
  python bench.py $B --steps 1500 > $O/C2_${v}_$rep.json 2>> $O/err
  python bench.py --workload C5 $B --steps 200 > $O/C5_${v}_$rep.json 2>> $O/err
done
done
unset MDH_LIB
timeout 600 python -m pytest tests/test_gpu_hough_accumulator.py tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -q 2>&1 | tail -30 > $O/pytest.log
cat $O/pytest.log

cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ae; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_configs.py -m gpu -q -x 2>&1 | tail -30 > $O/pytest.log
cat $O/pytest.log
B="--cpu-seconds 0 --parity-frames 0 --standalone-passes 8"
for rep in 1 2 3; do
for v in max3 nomax3 onebuf nodense; do
  unset MDH_LIB MDH_MORPH_DENSE
  if [ $v = nomax3 ]; then export MDH_LIB=$GRAFT_REPO_ROOT/metdetpy_amd/libmetdet_hip_nomax3.so; fi
  if [ $v = onebuf ]; then export MDH_LIB=$GRAFT_REPO_ROOT/metdetpy_amd/libmetdet_hip_onebuf.so; fi
  if [ $v = nodense ]; then export MDH_MORPH_DENSE=0; fi
  python bench.py $B --steps 1500 > $O/C2_${v}_$rep.json 2>> $O/err
  if [ $rep != 3 ]; then
  python bench.py --workload C3 $B --steps 300 > $O/C3_${v}_$rep.json 2>> $O/err
  python bench.py --workload C5 $B --steps 150 > $O/C5_${v}_$rep.json 2>> $O/err
  fi
done
done
unset MDH_LIB MDH_MORPH_DENSE
python tools/bench_per_frame.py > $O/per_frame.json 2>> $O/err

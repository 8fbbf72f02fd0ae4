cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ah; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hough_accumulator.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -12 > $O/pytest.log
cat $O/pytest.log
B="--cpu-seconds 0 --parity-frames 0 --standalone-passes 8"
run() { python bench.py $2 $B --steps $3 > $O/$1.json 2>> $O/err; }
for rep in 1 2; do
  run C2_default_$rep "" 1500
  MDH_WIN_BLOCK=64 run C2_blk64_$rep "" 1500
  MDH_WIN_BLOCK=128 run C2_blk128_$rep "" 1500
  run C2_b4096_$rep "--batch 4096" 800
  run C3_default_$rep "--workload C3" 300
  MDH_WIN_BLOCK=64 run C3_blk64_$rep "--workload C3" 300
  MDH_WIN_BLOCK=128 run C3_blk128_$rep "--workload C3" 300
done
tail -5 $O/err

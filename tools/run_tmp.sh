cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ak; mkdir -p $O
B="--cpu-seconds 0 --parity-frames 0 --standalone-passes 8"
run() { python bench.py $2 $B --steps $3 > $O/$1.json 2>> $O/err; }
for rep in 1 2; do
  run C5_g2048_$rep "--workload C5 --unique 50" 150
  MDH_DENSE_GRID=8192 run C5_g8192_$rep "--workload C5 --unique 50" 150
  MDH_DENSE_GRID=16384 run C5_g16384_$rep "--workload C5 --unique 50" 150
  MDH_DENSE_GRID=4096 run C5_g4096_$rep "--workload C5 --unique 50" 150
done
MDH_DENSE_GRID=8192 python bench.py --mode bands --workload C5 --unique 50 --cpu-seconds 0 > $O/bands_g8192.json 2>> $O/err
tail -3 $O/err

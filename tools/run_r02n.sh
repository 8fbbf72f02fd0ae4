cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
B="--cpu-seconds 0 --parity-frames 0 --standalone-passes 6 --steps 1500"
for rep in 1 2; do
python bench.py $B > $O/base_$rep.json 2> $O/err
for w in 8192 16384 32768; do
MDH_MAIN_STREAMS=1 MDH_WIN_WAVES=$w python bench.py $B > $O/m1_w${w}_$rep.json 2>> $O/err
MDH_WIN_WAVES=$w python bench.py $B > $O/m2_w${w}_$rep.json 2>> $O/err
done
MDH_MAIN_STREAMS=1 python bench.py $B > $O/m1_w4096_$rep.json 2>> $O/err
done

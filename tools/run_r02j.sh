cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j; mkdir -p $O
B="--cpu-seconds 0 --parity-frames 0 --standalone-passes 0 --steps 1500"
for rep in 1 2; do
python bench.py $B > $O/C2_default_$rep.json 2> $O/err
MDH_HOUGH_SETS=1 python bench.py $B > $O/C2_sets1_$rep.json 2>> $O/err
MDH_SLOTS=4 python bench.py $B > $O/C2_slots4_$rep.json 2>> $O/err
MDH_HOUGH_SETS=1 MDH_SLOTS=4 python bench.py $B > $O/C2_sets1_slots4_$rep.json 2>> $O/err
MDH_HOUGH_SETS=3 python bench.py $B > $O/C2_sets3_$rep.json 2>> $O/err
MDH_HOUGH_GRID_SMALL=512 python bench.py $B > $O/C2_g512_$rep.json 2>> $O/err
MDH_LIB=$PWD/metdetpy_amd/libmetdet_hip_prio0.so python bench.py $B > $O/C2_prio0_$rep.json 2>> $O/err
python bench.py --batch 1024 $B > $O/C2_b1024_$rep.json 2>> $O/err
python bench.py --batch 4096 --steps 700 --cpu-seconds 0 --parity-frames 0 --standalone-passes 0 > $O/C2_b4096_$rep.json 2>> $O/err
done
python bench.py --mode timeshard --cpu-seconds 0 > $O/timeshard_C2.json 2> $O/timeshard_C2.err

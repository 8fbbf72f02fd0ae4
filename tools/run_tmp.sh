cd $GRAFT_REPO_ROOT
TAG=r02d
O=gpurun_out/collect_$TAG; mkdir -p $O
python bench.py --workload C5 --unique 50 --cpu-seconds 6 > $O/${TAG}_bench_C5_8k_single_gpu.json 2> $O/err_C5
python bench.py --workload C5 --unique 50 --batch 128 --cpu-seconds 0 > $O/${TAG}_bench_C5_batch128.json 2>> $O/err_C5
python bench.py --mode bands --workload C5 --unique 50 --batch 256 --cpu-seconds 0 > $O/${TAG}_bench_mode_bands_C5_8bands_one_gpu.json 2> $O/err_bands
python bench.py --workload C5 --unique 50 --batch 128 --hough-mode accumulator --cpu-seconds 0 > $O/${TAG}_bench_C5_hough_accumulator.json 2>> $O/err_acc
python bench.py --protocol per-frame --cpu-seconds 0 > $O/${TAG}_bench_per_frame_protocol.json 2> $O/err_pf
python tools/bench_per_frame.py > $O/${TAG}_per_frame_protocol.json 2>> $O/err_pf
WL=C5 python tools/hough_frame_times.py > $O/${TAG}_hough_frame_times_C5.txt 2>&1
tail -2 $O/err_*

#!/bin/bash
# Everything profiles/ holds for a round, in one go on the GPU box:  bash tools/collect_round.sh <tag>   (outputs under gpurun_out/collect_<tag>/)
set -u
cd ${GRAFT_REPO_ROOT:-.}
TAG=${1:-r06}
O=gpurun_out/collect_$TAG; mkdir -p $O
S="--cpu-seconds 6 --no-secondary"
NS="--no-secondary --cpu-seconds 0"
# the headline line as the driver runs it (secondary block included: the other workloads and the per-frame protocol as subprocesses)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/err_default
python bench.py --workload C3 $S > $O/${TAG}_bench_C3_4k_n16_mask.json 2> $O/err_C3
python bench.py --workload C3 --batch 500 $S > $O/${TAG}_bench_C3_batch500.json 2>> $O/err_C3
python bench.py --workload C1s $S > $O/${TAG}_bench_C1s_960x540.json 2> $O/err_C1s
python bench.py --workload C5 --unique 50 $S > $O/${TAG}_bench_C5_8k_single_gpu.json 2> $O/err_C5
python bench.py --workload C5 --unique 50 --batch 125 $NS > $O/${TAG}_bench_C5_batch125.json 2>> $O/err_C5
python bench.py --batch 2000 $NS > $O/${TAG}_bench_C2_batch2000.json 2> $O/err_b2000
python bench.py --batch 8000 $NS > $O/${TAG}_bench_C2_batch8000.json 2> $O/err_b8000
python bench.py --batch 1000 $NS > $O/${TAG}_bench_C2_batch1000.json 2> $O/err_b1000
python bench.py --workload C3 --batch 2000 $NS > $O/${TAG}_bench_C3_batch2000.json 2>> $O/err_C3
python bench.py --batch 250 $NS > $O/${TAG}_bench_C2_batch250.json 2> $O/err_b250
python bench.py --host-input $NS --passes-per-step 1 --steps 200 > $O/${TAG}_bench_C2_host_input.json 2> $O/err_host
python bench.py --protocol per-frame --cpu-seconds 0 > $O/${TAG}_bench_per_frame_protocol.json 2> $O/err_pf
python bench.py --protocol per-frame --frames recycled --cpu-seconds 0 > $O/${TAG}_bench_per_frame_recycled_ring.json 2>> $O/err_pf
python bench.py --protocol per-frame --frames pageable --steps 6000 --cpu-seconds 0 > $O/${TAG}_bench_per_frame_pageable.json 2>> $O/err_pf
python tools/bench_per_frame.py > $O/${TAG}_per_frame_protocol.json 2>> $O/err_pf
# kernel trace of the per-frame loop (the window kernel launched ahead of its frame shows up as k_stack_threshold_mail)
(export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats -d $O/pf_stats -o r --output-format csv -- python bench.py --protocol per-frame --steps 3000 --cpu-seconds 0 --parity-frames 0 > $O/pf_stats.log 2>&1; cp $(find $O/pf_stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_rocprofv3_kernel_stats_per_frame.csv 2>/dev/null)
timeout 600 python bench.py --mode bands --workload C5 --unique 50 --batch 500 --steps 16 --warmup 3 --cpu-seconds 0 > $O/${TAG}_bench_mode_bands_C5_8bands_one_gpu.json 2> $O/err_bands
python bench.py --mode timeshard --cpu-seconds 0 > $O/${TAG}_bench_mode_timeshard_C2_8windows_one_gpu.json 2> $O/err_ts
python bench.py --mode timeshard --host-input --steps 16 --cpu-seconds 0 > $O/${TAG}_bench_mode_timeshard_C2_host_input.json 2>> $O/err_ts
python bench.py --detector classic $NS > $O/${TAG}_bench_classic_detector.json 2> $O/err_classic
MDH_CLASSIC_SPARSE=0 python bench.py --detector classic $NS --no-pmc > $O/${TAG}_bench_classic_detector_dense_kernels.json 2>> $O/err_classic
# the reference's default window (window_sec 1 -> n = 25 / 30) on the 8-pixels-per-lane window kernels, and round 1's byte-load kernel beside them
python bench.py --window 25 --unique 100 $NS > $O/${TAG}_bench_C2_n25.json 2> $O/err_n25
python bench.py --window 30 --unique 100 $NS --no-pmc > $O/${TAG}_bench_C2_n30.json 2>> $O/err_n25
python bench.py --window 17 --unique 100 $NS --no-pmc > $O/${TAG}_bench_C2_n17.json 2>> $O/err_n25
python bench.py --window 32 --unique 100 $NS --no-pmc > $O/${TAG}_bench_C2_n32.json 2>> $O/err_n25
python bench.py --workload C1s --window 25 --unique 100 $NS > $O/${TAG}_bench_C1s_n25.json 2>> $O/err_n25
python bench.py --workload C1s --window 30 --unique 100 $NS --no-pmc > $O/${TAG}_bench_C1s_n30.json 2>> $O/err_n25
MDH_WIDE8_MAXN=16 python bench.py --window 25 --unique 100 $NS --no-pmc --steps 8 > $O/${TAG}_bench_C2_n25_round1_byte_kernel.json 2>> $O/err_n25
MDH_SKIP_HOUGH=1 python bench.py --window 25 --unique 100 $NS --no-pmc --parity-frames 0 --steps 8 > $O/${TAG}_bench_C2_n25_without_hough_stage.json 2>> $O/err_n25
python bench.py --with-collector $NS --no-pmc > $O/${TAG}_bench_C2_with_collector.json 2> $O/err_col2
python bench.py --hough-mode accumulator $NS --steps 6 > $O/${TAG}_bench_C2_hough_accumulator.json 2> $O/err_acc
python bench.py --workload C3 --hough-mode accumulator $NS --steps 6 > $O/${TAG}_bench_C3_hough_accumulator.json 2>> $O/err_acc
python bench.py --workload C5 --unique 50 --batch 125 --hough-mode accumulator $NS --steps 6 > $O/${TAG}_bench_C5_hough_accumulator.json 2>> $O/err_acc
python bench.py --workload C1e --cpu-seconds 0 > $O/${TAG}_bench_C1e_end_to_end.json 2> $O/err_c1e
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 1 --cpu-seconds 0 2> $O/err_2r | grep '^{' > $O/${TAG}_bench_2ranks_one_gpu.json
# the unfriendly skies on their own lines (they are also legs of the headline run's `secondary` block)
for st in noisy hot cloud sparse dense; do
  U=100; [ $st = sparse ] && U=400; [ $st = dense ] && U=96; BB=""; [ $st = dense ] && BB="--batch 3840"
  python bench.py --stream $st --unique $U $BB $NS --no-pmc --steps 6 --warmup 1 > $O/${TAG}_bench_C2_stream_$st.json 2>> $O/err_streams
done
bash tools/run_8gpu.sh --dry-run 8 $O/dry8 > $O/${TAG}_run_8gpu_dry_run.txt 2>&1
for wl in C2 C3 C5 C1s; do WL=$wl python tools/hough_frame_times.py > $O/${TAG}_hough_frame_times_$wl.txt 2>&1; done
./tools/lds_atomic_rate > $O/${TAG}_lds_atomic_rate.txt 2>&1
python tools/bench_pre_batch.py > $O/${TAG}_preprocess_batch.json 2> $O/err_pre
python tools/bench_collector.py > $O/${TAG}_collector.json 2> $O/err_col
bash tools/profile_round.sh $TAG C2 > $O/profile_C2.log 2>&1
NAME=C2n25 EXTRA="--window 25 --unique 100" bash tools/profile_round.sh $TAG C2 > $O/profile_C2n25.log 2>&1
NAME=C1sn25 EXTRA="--window 25 --unique 100" bash tools/profile_round.sh $TAG C1s > $O/profile_C1sn25.log 2>&1
NAME=classic EXTRA="--detector classic" bash tools/profile_round.sh $TAG C2 > $O/profile_classic.log 2>&1
bash tools/pmc_pre.sh $O/pmc_pre > $O/${TAG}_pmc_preprocess.txt 2>&1; rm -rf $O/pmc_pre
MDH_PROBE_N=25 bash tools/pmc_window_frontend.sh $O/pmc_n25 > $O/${TAG}_pmc_window_n25.txt 2>&1; rm -rf $O/pmc_n25
MDH_PROBE_N=8 bash tools/pmc_window_frontend.sh $O/pmc_n8 > $O/${TAG}_pmc_window_n8.txt 2>&1; rm -rf $O/pmc_n8
./tools/load_width_probe > $O/${TAG}_load_width_probe.txt 2>&1
for mode in pre detector; do timeout 200 python tools/stress_pre_run.py --mode $mode --iters 1000 --seconds 100 > $O/${TAG}_stress_${mode}.txt 2>&1; echo "exit code $?" >> $O/${TAG}_stress_${mode}.txt; done
bash tools/profile_round.sh $TAG C3 > $O/profile_C3.log 2>&1
bash tools/profile_round.sh $TAG C1s > $O/profile_C1s.log 2>&1
EXTRA="--unique 50" bash tools/profile_round.sh $TAG C5 > $O/profile_C5.log 2>&1
MDH_DENSE=0 python bench.py --workload C5 --unique 50 $NS --no-pmc --steps 8 > $O/${TAG}_bench_C5_sparse_format.json 2>> $O/err_C5
cp gpurun_out/profile_${TAG}_C2/${TAG}_rocprofv3_kernel_stats_C2.csv gpurun_out/profile_${TAG}_C2/${TAG}_rocprofv3_kernel_stats_C2_depth1.csv gpurun_out/profile_${TAG}_C2/pmc_traffic_C2.json $O/ 2>/dev/null
cp gpurun_out/profile_${TAG}_C5/${TAG}_rocprofv3_kernel_stats_C5.csv gpurun_out/profile_${TAG}_C5/${TAG}_rocprofv3_kernel_stats_C5_depth1.csv gpurun_out/profile_${TAG}_C5/pmc_traffic_C5.json $O/ 2>/dev/null
cp gpurun_out/profile_${TAG}_C1s/${TAG}_rocprofv3_kernel_stats_C1s.csv gpurun_out/profile_${TAG}_C1s/${TAG}_rocprofv3_kernel_stats_C1s_depth1.csv gpurun_out/profile_${TAG}_C1s/pmc_traffic_C1s.json $O/ 2>/dev/null
for nm in C2n25 C1sn25 classic; do cp gpurun_out/profile_${TAG}_$nm/${TAG}_rocprofv3_kernel_stats_$nm.csv gpurun_out/profile_${TAG}_$nm/${TAG}_rocprofv3_kernel_stats_${nm}_depth1.csv gpurun_out/profile_${TAG}_$nm/pmc_traffic_$nm.json $O/ 2>/dev/null; done
cp gpurun_out/profile_${TAG}_C3/${TAG}_rocprofv3_kernel_stats_C3.csv gpurun_out/profile_${TAG}_C3/${TAG}_rocprofv3_kernel_stats_C3_depth1.csv gpurun_out/profile_${TAG}_C3/pmc_traffic_C3.json $O/ 2>/dev/null
ls -la $O | head -60

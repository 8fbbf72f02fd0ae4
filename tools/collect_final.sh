#!/bin/bash
# The subset of tools/collect_round.sh that the FINAL library of a round changes (round 6: Hough band routing and the register pins came
# after the round's main collection): the headline line with its secondary block, the long-window / ClassicDetector / 4K / 960x540 legs
# on their own, and the rocprofv3 profiles the README's table is made from.  bash tools/collect_final.sh <tag>  (outputs under gpurun_out/collect_<tag>_final/)
set -u
cd ${GRAFT_REPO_ROOT:-.}
TAG=${1:-r06}
O=gpurun_out/collect_${TAG}_final; mkdir -p $O
NS="--no-secondary --cpu-seconds 0"
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_default.json 2> $O/err_default
python bench.py --window 25 --unique 100 $NS > $O/${TAG}_bench_C2_n25.json 2> $O/err_n25
python bench.py --window 30 --unique 100 $NS --no-pmc > $O/${TAG}_bench_C2_n30.json 2>> $O/err_n25
python bench.py --workload C1s --window 25 --unique 100 $NS --no-pmc > $O/${TAG}_bench_C1s_n25.json 2>> $O/err_n25
MDH_HOUGH_BANDS=0 python bench.py --window 25 --unique 100 $NS --no-pmc > $O/${TAG}_bench_C2_n25_corner_routing_only.json 2>> $O/err_n25
python bench.py --detector classic $NS > $O/${TAG}_bench_classic_detector.json 2> $O/err_classic
python bench.py --workload C3 --cpu-seconds 0 --no-secondary > $O/${TAG}_bench_C3_4k_n16_mask.json 2> $O/err_C3
python bench.py --workload C1s --cpu-seconds 0 --no-secondary > $O/${TAG}_bench_C1s_960x540.json 2> $O/err_C1s
python bench.py --with-collector $NS --no-pmc > $O/${TAG}_bench_C2_with_collector.json 2> $O/err_col2
for wl in C2 C1s; do WL=$wl WINDOW=25 python tools/hough_frame_times.py > $O/${TAG}_hough_frame_times_${wl}_n25.txt 2>&1; done
bash tools/profile_round.sh $TAG C2 > $O/profile_C2.log 2>&1
NAME=C2n25 EXTRA="--window 25 --unique 100" bash tools/profile_round.sh $TAG C2 > $O/profile_C2n25.log 2>&1
NAME=classic EXTRA="--detector classic" bash tools/profile_round.sh $TAG C2 > $O/profile_classic.log 2>&1
bash tools/profile_round.sh $TAG C3 > $O/profile_C3.log 2>&1
bash tools/profile_round.sh $TAG C1s > $O/profile_C1s.log 2>&1
for nm in C2 C2n25 classic C3 C1s; do cp gpurun_out/profile_${TAG}_$nm/${TAG}_rocprofv3_kernel_stats_$nm.csv gpurun_out/profile_${TAG}_$nm/${TAG}_rocprofv3_kernel_stats_${nm}_depth1.csv gpurun_out/profile_${TAG}_$nm/pmc_traffic_$nm.json $O/ 2>/dev/null; done
ls -la $O | head -40

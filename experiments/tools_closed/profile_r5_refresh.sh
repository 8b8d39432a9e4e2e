#!/bin/bash
# Round 5, after the sixteen-tile level (kz_tower_f16g.hip / kz_tower_pairs_shapes.hpp changed): the traffic records of the pair
# towers again, the bench lines and the kernel stats of the default command.  Run on the GPU box from the repo root.
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
STEPS=30 WARMUP=5 bash tools/pmc_traffic.sh chess-20x256 f32split16
STEPS=60 WARMUP=10 bash tools/pmc_traffic.sh ataxx-8x128 f32split16
STEPS=20 WARMUP=5 bash tools/pmc_traffic.sh go9-16x128 f32split16
STEPS=20 WARMUP=5 bash tools/pmc_traffic.sh go9-16x128 f16
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
rm -rf $O/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 2000 --repeats 3 --no-cpu-baseline --no-seam > $O/stats_bench.json 2> $O/stats.log
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats_default_bench.csv
rm -rf $O/stats
bash tools/pmc_workload.sh go9-16x128 f16 20 2>&1 | grep -v "^$" > $O/mfma_busy_go9_tiles16.txt
python3 tools/show_bench.py $O/bench.json | cut -c1-300
mkdir -p $O/pmc
for d in gpurun_out/pmc_traffic_*_FETCH_SIZE gpurun_out/pmc_traffic_*_WRITE_SIZE; do
  f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/pmc/$(basename $d).csv
done
cp profiles/hbm_traffic.json $O/hbm_traffic.json

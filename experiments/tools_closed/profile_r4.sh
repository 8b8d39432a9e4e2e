#!/bin/bash
# Round-4 evidence, run on the GPU box from the repo root: the default bench line, the driver's command, rocprofv3 kernel
# stats of the default command (sub-records included, the seam left out: a profiled process must not start GPU-using
# children), FETCH_SIZE / WRITE_SIZE passes of every dominant kernel (all five kernel sources changed this round), the
# shape-generality sweep (path, deviation from the oracle, evals/s at depth 20) and the executor bench with the real host
# work.  Copies what is judged into profiles/r4/ (tracked).
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O profiles/r4
STEPS=30 WARMUP=5 bash tools/pmc_traffic.sh chess-20x256 f16
STEPS=4 WARMUP=1 bash tools/pmc_traffic.sh go19-40x256 f16
STEPS=30 WARMUP=5 bash tools/pmc_traffic.sh chess-20x256 f32split16
STEPS=60 WARMUP=10 bash tools/pmc_traffic.sh ataxx-8x128 f32
STEPS=60 WARMUP=10 bash tools/pmc_traffic.sh ataxx-8x128 f32split16
STEPS=3 WARMUP=1 bash tools/pmc_traffic.sh go19-40x256 f32split16
STEPS=20 WARMUP=5 bash tools/pmc_traffic.sh go9-16x128 f32split16
STEPS=20 WARMUP=5 bash tools/pmc_traffic.sh go9-16x128 f16
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
rm -rf $O/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 2000 --no-cpu-baseline --no-seam > $O/stats_bench.json 2> $O/stats.log
f=$(ls -S $(find $O/stats -name "*kernel_stats.csv") | head -1)
cp "$f" $O/kernel_stats_default_bench.csv
rm -rf $O/stats
python3 tools/show_bench.py $O/bench.json
python3 tools/show_bench.py $O/bench_driver_cmd.json
head -14 $O/kernel_stats_default_bench.csv | cut -c1-170
python3 tools/shape_sweep.py --out $O/shape_sweep.json > $O/shape_sweep.log 2>&1
bash tools/bench_executor_r4.sh 3 > $O/bench_executor.jsonl 2> $O/bench_executor.err
cp $O/bench.json $O/bench_driver_cmd.json $O/kernel_stats_default_bench.csv $O/shape_sweep.json $O/bench_executor.jsonl profiles/r4/
mkdir -p $O/pmc
for d in gpurun_out/pmc_traffic_*_FETCH_SIZE gpurun_out/pmc_traffic_*_WRITE_SIZE; do
  f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/pmc/$(basename $d).csv
done
cp profiles/hbm_traffic.json $O/hbm_traffic.json

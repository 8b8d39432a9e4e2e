#!/bin/bash
# Round-3 evidence, run on the GPU box from the repo root: the default bench line, the driver's command, rocprofv3 kernel
# stats of the default command (sub-records included, the seam left out: a profiled process must not start GPU-using
# children), FETCH_SIZE / WRITE_SIZE passes of the kernels that changed this round (Go board convolution, split-f16
# launch), clock / MFMA-busy pass of the Go kernel.  Copies what is judged into profiles/r3/ (tracked).
export TMPDIR=/tmp
O=gpurun_out/r3
mkdir -p $O profiles/r3
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
rm -rf $O/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 2000 --no-cpu-baseline --no-seam > $O/stats_bench.json 2> $O/stats.log
# the stats file of the bench process itself (a run directory may hold more than one process's files)
f=$(ls -S $(find $O/stats -name "*kernel_stats.csv") | head -1)
cp "$f" $O/kernel_stats_default_bench.csv
rm -rf $O/stats
python3 tools/show_bench.py $O/bench.json
python3 tools/show_bench.py $O/bench_driver_cmd.json
head -12 $O/kernel_stats_default_bench.csv | cut -c1-170
STEPS=4 WARMUP=1 bash tools/pmc_traffic.sh go19-40x256 f16
STEPS=30 WARMUP=5 bash tools/pmc_traffic.sh chess-20x256 f32split16
STEPS=60 WARMUP=10 bash tools/pmc_traffic.sh ataxx-8x128 f32
STEPS=60 WARMUP=10 bash tools/pmc_traffic.sh ataxx-8x128 f32split16
STEPS=3 WARMUP=1 bash tools/pmc_traffic.sh go19-40x256 f32split16
bash tools/pmc_go.sh | tee $O/clock_and_mfma_go19.txt
[ -f kzero_amd/libkzhip_clock.so ] && bash tools/go_clock.sh && cp gpurun_out/goclock/clock.txt $O/go_clock.txt
cp $O/bench.json profiles/r3/bench.json
cp $O/bench_driver_cmd.json profiles/r3/bench_driver_cmd.json
cp $O/kernel_stats_default_bench.csv profiles/r3/kernel_stats_default_bench.csv
cp $O/clock_and_mfma_go19.txt profiles/r3/clock_and_mfma_go19.txt
cp profiles/hbm_traffic.json $O/hbm_traffic.json
mkdir -p $O/pmc
for d in gpurun_out/pmc_traffic_ataxx-8x128_f32_FETCH_SIZE gpurun_out/pmc_traffic_ataxx-8x128_f32_WRITE_SIZE gpurun_out/pmc_traffic_ataxx-8x128_f32split16_FETCH_SIZE gpurun_out/pmc_traffic_ataxx-8x128_f32split16_WRITE_SIZE gpurun_out/pmc_traffic_go19-40x256_f32split16_FETCH_SIZE gpurun_out/pmc_traffic_go19-40x256_f32split16_WRITE_SIZE gpurun_out/pmc_traffic_go19-40x256_f16_FETCH_SIZE gpurun_out/pmc_traffic_go19-40x256_f16_WRITE_SIZE gpurun_out/pmc_traffic_chess-20x256_f32split16_FETCH_SIZE gpurun_out/pmc_traffic_chess-20x256_f32split16_WRITE_SIZE; do
  f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp "$f" $O/pmc/$(basename $d).csv
done
cp $(find gpurun_out/pmc_go_clk -name "*counter_collection.csv" | head -1) $O/pmc/clock_and_mfma_go19_counters.csv 2>/dev/null

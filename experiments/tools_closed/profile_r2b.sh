#!/bin/bash
# Round-2 evidence refresh (after the one-launch exact-f32 network), run on the GPU box from the repo root: the default
# bench line, the driver's command, rocprofv3 kernel stats of the default command and of the A1 sub-record's command.
export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
python3 bench.py > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2b/bench_driver_cmd.json 2> gpurun_out/r2b/bench_driver_cmd.err
rm -rf gpurun_out/r2b/stats gpurun_out/r2b/stats_a1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2b/stats -o run -- python3 bench.py --steps 2000 --no-cpu-baseline --no-seam > gpurun_out/r2b/stats_bench.json 2> gpurun_out/r2b/stats.log
cp $(find gpurun_out/r2b/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r2b/kernel_stats_default_bench.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2b/stats_a1 -o run -- python3 bench.py --workload ataxx-8x128 --dtype f32 --steps 3000 --no-cpu-baseline --no-others --no-seam > gpurun_out/r2b/a1_bench.json 2> gpurun_out/r2b/stats_a1.log
cp $(find gpurun_out/r2b/stats_a1 -name "*kernel_stats.csv" | head -1) gpurun_out/r2b/kernel_stats_a1_f32.csv
rm -rf gpurun_out/r2b/stats gpurun_out/r2b/stats_a1
python3 tools/show_bench.py gpurun_out/r2b/bench.json
python3 tools/show_bench.py gpurun_out/r2b/bench_driver_cmd.json
python3 tools/show_bench.py gpurun_out/r2b/a1_bench.json
head -8 gpurun_out/r2b/kernel_stats_default_bench.csv | cut -c1-170
head -5 gpurun_out/r2b/kernel_stats_a1_f32.csv | cut -c1-170

#!/bin/bash
# Round-2 evidence, run on the GPU box from the repo root: the default bench line, rocprofv3 kernel stats of the same
# command, and the HBM-traffic PMC passes of the dominant kernels (separate passes, never combined with traces).
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
python3 bench.py > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err
rm -rf gpurun_out/r2/stats
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2/stats -o run -- python3 bench.py --steps 2000 --no-cpu-baseline --no-seam > gpurun_out/r2/stats_bench.json 2> gpurun_out/r2/stats.log
cp $(find gpurun_out/r2/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r2/kernel_stats.csv
bash tools/pmc_traffic.sh chess-20x256 f16 > gpurun_out/r2/traffic_chess.log 2>&1
bash tools/pmc_traffic.sh go19-40x256 f16 > gpurun_out/r2/traffic_go.log 2>&1
cp profiles/hbm_traffic.json gpurun_out/r2/hbm_traffic.json
python3 tools/show_bench.py gpurun_out/r2/bench.json
head -12 gpurun_out/r2/kernel_stats.csv | cut -c1-160
tail -2 gpurun_out/r2/traffic_chess.log; tail -2 gpurun_out/r2/traffic_go.log

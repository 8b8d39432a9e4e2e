#!/bin/bash
# Round 5, final tree: the whole GPU suite, the traffic records whose kernel sources changed since they were taken (Go 19x19: the
# priority-scheme macro) and the AttentionTower kernels', the bench lines, the kernel stats of the default command.  Run on the
# GPU box from the repo root; results under gpurun_out/r5f/.
export TMPDIR=/tmp
O=gpurun_out/r5f
mkdir -p $O
timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
timeout 300 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "attention_tower or att3x256" 2>&1 | grep "^\[f16\]" > $O/att_f16_deviation.txt
STEPS=4 WARMUP=1 bash tools/pmc_traffic.sh go19-40x256 f16
STEPS=4 WARMUP=1 bash tools/pmc_traffic.sh go19-40x256 f32split16
STEPS=40 WARMUP=10 bash tools/pmc_traffic.sh chess-att16x256 f16
STEPS=20 WARMUP=5 bash tools/pmc_traffic.sh chess-att16x256 f32
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
for dt in f16 f32; do
  python3 bench.py --workload chess-att16x256 --dtype $dt --no-others --no-cpu-baseline > $O/bench_att16x256_$dt.json 2> $O/bench_att_$dt.err
done
rm -rf $O/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --workload chess-att16x256 --dtype f32 --steps 300 --repeats 3 --no-others --no-cpu-baseline --no-seam > $O/stats_att_f32.json 2> $O/stats.log
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats_att16x256_f32.csv
rm -rf $O/stats
python3 tools/show_bench.py $O/bench.json | cut -c1-300
mkdir -p $O/pmc
for d in gpurun_out/pmc_traffic_*_FETCH_SIZE gpurun_out/pmc_traffic_*_WRITE_SIZE; do
  f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/pmc/$(basename $d).csv
done
cp profiles/hbm_traffic.json $O/hbm_traffic.json

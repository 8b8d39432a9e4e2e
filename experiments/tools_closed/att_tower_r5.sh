#!/bin/bash
# Round 5: the AttentionTower network (bench workload chess-att16x256) — parity tests, bench lines in f16 and exact f32, kernel
# stats of the f16 run.  Run on the GPU box from the repo root; results under gpurun_out/att/.
export TMPDIR=/tmp
O=gpurun_out/att
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "att2x64 or att2x32 or att3x256 or attention_tower" > $O/tests.log 2>&1
tail -3 $O/tests.log
for dt in ${ATT_DTYPES:-f16 f32}; do
  python3 bench.py --workload chess-att16x256 --dtype $dt --no-others --no-cpu-baseline ${ATT_BENCH_ARGS:-} > $O/bench_$dt.json 2> $O/bench_$dt.err
  python3 tools/show_bench.py $O/bench_$dt.json | cut -c1-400
done
rm -rf $O/stats
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --workload chess-att16x256 --dtype f16 --steps 500 --repeats 3 --no-others --no-cpu-baseline --no-seam > $O/stats_bench.json 2> $O/stats.log
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats_att_f16.csv && head -8 $O/kernel_stats_att_f16.csv | cut -c1-200
rm -rf $O/stats

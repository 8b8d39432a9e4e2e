#!/bin/bash
# Round-5 evidence, run on the GPU box from the repo root: FETCH_SIZE / WRITE_SIZE passes of every dominant kernel (every
# kernel source changed this round: the decode moved into the launches, the board-tile image was re-laid out, the pair tower
# was split into translation units), the default bench line, the driver's command, rocprofv3 kernel stats of the default
# command AND of the seam (bench_executor with the shim's default: there must be no kz_decode_output row), counter passes of
# the dominant kernels, the shape sweep and the executor bench.  Copies what is judged into profiles/r5/ (tracked).
export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O profiles/r5
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
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 bench.py --steps 2000 --repeats 3 --no-cpu-baseline --no-seam > $O/stats_bench.json 2> $O/stats.log
f=$(ls -S $(find $O/stats -name "*kernel_stats.csv") | head -1)
cp "$f" $O/kernel_stats_default_bench.csv
rm -rf $O/stats
# the seam as the Rust shim runs it by default (decode inside the launch, one prep helper), under the kernel trace: one
# launch per batch and nothing else
bash tools/bench_executor_r5.sh 1 > /dev/null 2>&1   # (builds the executable and the model file)
rm -rf $O/stats_seam
KZ_BENCH_CLEAN_EXIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_seam -o run -- tests/cpp/build/bench_executor /tmp/chess20x256.kzm 3 1 6 256 8 f16 3 1 0 real 1 > $O/stats_seam.json 2> $O/stats_seam.log
f=$(find $O/stats_seam -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats_seam_default.csv || echo "no kernel stats of the seam run" > $O/kernel_stats_seam_default.csv
rm -rf $O/stats_seam
python3 tools/show_bench.py $O/bench.json
python3 tools/show_bench.py $O/bench_driver_cmd.json | head -3
head -14 $O/kernel_stats_default_bench.csv | cut -c1-170
cat $O/kernel_stats_seam_default.csv | cut -c1-170
bash tools/bench_executor_r5.sh 3 > $O/bench_executor.jsonl 2> $O/bench_executor.err
: > $O/mfma_busy.txt
for w in "chess-20x256 f16 30" "chess-20x256 f32split16 20" "ataxx-8x128 f32 60" "go19-40x256 f16 3" "go19-40x256 f32split16 2" "go9-16x128 f16 20"; do
  set -- $w
  bash tools/pmc_workload.sh $1 $2 $3 2>&1 | grep -v "^$" >> $O/mfma_busy.txt
done
cat $O/mfma_busy.txt
python3 tools/shape_sweep.py --out $O/shape_sweep.json > $O/shape_sweep.log 2>&1
cp $O/bench.json $O/bench_driver_cmd.json $O/kernel_stats_default_bench.csv $O/kernel_stats_seam_default.csv $O/shape_sweep.json $O/bench_executor.jsonl $O/mfma_busy.txt profiles/r5/
mkdir -p $O/pmc
for d in gpurun_out/pmc_traffic_*_FETCH_SIZE gpurun_out/pmc_traffic_*_WRITE_SIZE; do
  f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/pmc/$(basename $d).csv
done
cp profiles/hbm_traffic.json $O/hbm_traffic.json

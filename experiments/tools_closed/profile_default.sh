export TMPDIR=/tmp
rm -rf gpurun_out/prof_r1b
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r1b -o run -- python3 bench.py > gpurun_out/prof_r1b_bench.json 2> gpurun_out/prof_r1b.log
python3 bench.py > gpurun_out/bench_r1b.json 2>gpurun_out/bench_r1b.err
for c in FETCH_SIZE WRITE_SIZE; do rm -rf gpurun_out/pmc_r1b_$c; rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_r1b_$c -o run -- python3 bench.py --no-cpu-baseline --engines 1 --steps 100 --warmup 10 > gpurun_out/pmc_r1b_$c.log 2>&1; done
cat gpurun_out/bench_r1b.json

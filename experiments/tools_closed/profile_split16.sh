# rocprofv3 kernel stats + clean bench lines of the split-f16 and one-launch-f16 towers (copied into profiles/ by hand)
export TMPDIR=/tmp
for cfg in "ataxx-8x128 f32split16 a1_split16" "chess-20x256 f32split16 chess_split16" "ataxx-8x128 f16 a1_f16g"; do
  set -- $cfg
  rm -rf gpurun_out/prof_$3
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$3 -o run -- python3 bench.py --workload $1 --dtype $2 --no-cpu-baseline > gpurun_out/prof_$3.json 2> gpurun_out/prof_$3.log
  python3 bench.py --workload $1 --dtype $2 --no-cpu-baseline > gpurun_out/bench_$3.json 2>/dev/null
  cat gpurun_out/bench_$3.json | cut -c1-400
done

bash tools/bench_executor_r5.sh 1 >/dev/null 2>&1
for h in 0 1 2; do
  tests/cpp/build/bench_executor /tmp/chess20x256.kzm 3 1 6 256 8 f16 3 1 0 real $h 2>&1 >/tmp/out.json | grep "executor 0"
  python3 -c "import json;d=json.load(open('/tmp/out.json'));print('helpers',d['prep_helpers'],round(d['evals_per_s']),d['executor_work_util'],d['helper_cpu_util'], 'work us/batch', round(d['executor_work_s_per_Meval']*256,1))"
done

export KZ_NO_FUSED_HEADS=1 KZ_TOWER_NB=${NB:-4}
# (experiment kernels: needs the experiment build, experiments/build.sh)
export KZ_LIB_PATH=${KZ_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/experiments/libkzhip_exp.so}
for e in 1 2 3 4 6; do
python bench.py --no-others --no-cpu-baseline --no-host-io --steps 1500 --warmup 30 --engines $e ${BENCH_ARGS:-} | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('NB=$KZ_TOWER_NB engines $e evals/s', r['value'], 'launch ms', r['roofline']['avg_launch_ms'], 'wgs', r['roofline']['workgroups_per_launch'])"
done

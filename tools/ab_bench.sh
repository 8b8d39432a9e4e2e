#!/bin/bash
# same-box A/B of library builds: tools/ab_bench.sh <rounds> <suffix...>   ("" = the product build libkzhip.so)
rounds=$1; shift
for r in $(seq $rounds); do
  for v in "$@"; do
    lib=/root/repo/kzero_amd/libkzhip$v.so
    KZ_LIB_PATH=$lib python bench.py --no-cpu-baseline --steps ${STEPS:-4000} ${BENCH_ARGS:-} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib$v', d['value'], d['roofline']['avg_launch_ms'])"
  done
done

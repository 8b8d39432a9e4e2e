#!/bin/bash
# same-box A/B of library builds on the 128-channel benchmark workloads: tools/ab_libs.sh <suffix...>  ("" = libkzhip.so)
LIBS="$*"; for r in 1 2; do for v in $LIBS; do for w in "go9-16x128 300" "ataxx-8x128 3000"; do set -- $w; for d in f16 f32split16; do
  KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so python bench.py --workload $1 --dtype $d --no-cpu-baseline --no-host-io --steps $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$1 $d lib$v\", d[\"value\"], d[\"roofline\"][\"avg_launch_ms\"])"
done; done; done; done

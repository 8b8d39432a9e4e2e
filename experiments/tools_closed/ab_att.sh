#!/bin/bash
# same-box A/B of library builds on the AttentionTower workload (chess-att16x256, f16), interleaved:
#   tools/ab_att.sh "" _nb1      (suffixes of kzero_amd/libkzhip<suffix>.so; "" = the working tree's library)
for r in 1 2 3; do for v in "$@"; do
  KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so python bench.py --workload chess-att16x256 --dtype f16 --no-cpu-baseline --no-host-io --no-others --no-seam --repeats 3 --steps ${STEPS:-1000} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"chess-att16x256 f16 lib$v\", d[\"value\"], d[\"value_min\"], d[\"value_max\"], d[\"roofline\"][\"avg_launch_ms\"])"
done; done

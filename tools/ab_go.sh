#!/bin/bash
# same-box A/B of library builds on the Go 19x19 40x256 b=512 workload (BASELINE configs[4]), f16 and split16, interleaved:
#   tools/ab_go.sh _prev ""        (suffixes of kzero_amd/libkzhip<suffix>.so; "" = the working tree's library)
for r in 1 2 3; do for v in "$@"; do for d in f16 f32split16; do
  KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so python bench.py --workload go19-40x256 --dtype $d --no-cpu-baseline --boundary resident --no-host-io --no-others --no-seam --repeats 3 --steps ${STEPS:-200} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"go19-40x256 $d lib$v\", d[\"value\"], d[\"value_min\"], d[\"value_max\"], d[\"roofline\"][\"avg_launch_ms\"])"
done; done; done

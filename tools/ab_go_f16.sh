#!/bin/bash
# same-box A/B of library builds on Go 19x19 40x256 b=512 f16 (device-resident: the kernel alone), interleaved, three rounds:
#   tools/ab_go_f16.sh _prev "" _ntw2        (suffixes of kzero_amd/libkzhip<suffix>.so; "" = the working tree's library)
for r in 1 2 3; do for v in "$@"; do
  KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so python3 bench.py --workload go19-40x256 --dtype f16 --no-cpu-baseline --boundary resident --no-host-io --no-others --no-seam --repeats 3 --steps ${STEPS:-200} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('go19-40x256 f16 lib$v', d['value'], d['value_min'], d['value_max'], d['roofline']['avg_launch_ms'])"
done; done

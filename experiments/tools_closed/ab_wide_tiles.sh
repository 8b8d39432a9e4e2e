#!/bin/bash
# same-box A/B of the wide tiles of the plain-f16 one-launch tower (twice the boards per workgroup at 128 channels) against
# a library built before them (tools/build_rev_lib.sh <rev> prewide): deep and shallow networks, batches 256 .. 2048
for cfg in "256 20" "1024 20" "256 8" "512 8" "1024 8"; do set -- $cfg; echo "### batch $1, depth $2"
  BATCH=$1 RATE_DEPTH=$2 FILTER=chess_3x128_att,go-9-noterr_3x128_conv,ataxx-7_3x128_ataxx_conv,chess_3x96 DTYPES=f16 bash tools/ab_sweep_libs.sh . _prewide
done
for r in 1 2; do for v in "" _prewide; do KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so python bench.py --workload go9-16x128 --dtype f16 --no-cpu-baseline --no-host-io --no-others --no-seam --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('go9-16x128 f16 b2048 lib$v', d['value'], d['roofline']['frac'], d['config']['tower_path'])"; done; done

#!/bin/bash
# same-box A/B of the plain-f16 launch's own heads tail (f16 MFMAs on the f16 images; fused heads with the wide tiles too)
# against a library built before it (tools/build_rev_lib.sh <rev> prev)
for cfg in "256 20" "1024 20" "256 8" "1024 8"; do set -- $cfg; echo "### batch $1, depth $2"
  BATCH=$1 RATE_DEPTH=$2 FILTER=go-9-noterr_3x128_conv,ataxx-7_3x128_ataxx_conv,go-9_3x128_conv_scalar,ataxx-8_3x256 DTYPES=f16 bash tools/ab_sweep_libs.sh . _prev
done
for r in 1 2; do for v in "" _prev; do for w in "go9-16x128 300" "ataxx-8x128 3000"; do set -- $w
  KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so python bench.py --workload $1 --dtype f16 --no-cpu-baseline --no-host-io --no-others --no-seam --steps $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 f16 lib$v', d['value'], d['roofline']['frac'], d['config']['tower_path'])"
done; done; done

#!/bin/bash
# same-box A/B of library builds on sweep cases (rate at depth 20, batch 256): tools/ab_sweep_libs.sh <suffix...>
# ("." = libkzhip.so, "_x" = libkzhip_x.so built by tools/build_rev_lib.sh); FILTER = comma-separated case substrings
# (default: the conv-head shapes that run the fused tail of kz_conv_heads.hpp), DTYPES = f32,f16,parity, BATCH
LIBS="$*"
FILTER=${FILTER:-go-9_3x128_conv,ataxx-7_3x128_ataxx_conv,go-9-noterr_3x128}
DTYPES=${DTYPES:-f32,f16,parity}
BATCH=${BATCH:-0}  # 0 = the sweep's own (256; 128 on Go 19x19)
RATE_DEPTH=${RATE_DEPTH:-20}
for r in 1 2; do for v in $LIBS; do
  [ "$v" = "." ] && v=""
  echo "== lib$v run $r"
  KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so python tools/shape_sweep.py --no-oracle --filter "$FILTER" --dtypes $DTYPES --batch $BATCH --rate-depth $RATE_DEPTH --out gpurun_out/ab_tail$v$r.json > /dev/null 2>&1
  python tools/show_sweep.py gpurun_out/ab_tail$v$r.json
done; done

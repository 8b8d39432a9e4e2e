#!/bin/bash
# same-box A/B of the fused heads tail (kz_conv_heads.hpp) on the head shapes that go through it:
# tools/ab_heads_tail.sh <suffix...>  ("." = libkzhip.so); prints the sweep rows (rate at depth 20, batch 256)
LIBS="$*"
for r in 1 2; do for v in $LIBS; do
  [ "$v" = "." ] && v=""
  echo "== lib$v run $r"
  KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so python tools/shape_sweep.py --no-oracle --filter "go-9_3x128_conv,ataxx-7_3x128_ataxx_conv,go-9-noterr_3x128" --out gpurun_out/ab_tail$v$r.json > /dev/null 2>&1
  python tools/show_sweep.py gpurun_out/ab_tail$v$r.json
done; done

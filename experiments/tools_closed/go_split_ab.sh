#!/bin/bash
# Go-19 40x256 B=512 at KZ_DTYPE_F32_SPLIT16 through the product library ("new") and alternative builds
# kzero_amd/libkzhip_<name>.so (tools/build_rev_lib.sh), alternating, plus the split board-conv parity tests.
# Usage (GPU box): LIBS="cur new" bash tools/go_split_ab.sh
mkdir -p gpurun_out/gosab
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "board_conv_split16 or go19_40x256_split16" 2>&1 | tail -2
for rep in 1 2; do for lib in ${LIBS:-cur new}; do
  if [ $lib != new ]; then [ -f kzero_amd/libkzhip_$lib.so ] || continue; export KZ_LIB_PATH=$PWD/kzero_amd/libkzhip_$lib.so; else unset KZ_LIB_PATH; fi
  python bench.py --workload go19-40x256 --dtype f32split16 --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-others --no-seam --no-host-io > gpurun_out/gosab/$lib$rep.json 2> gpurun_out/gosab/$lib$rep.err
  python -c "
import json
r=json.loads(open('gpurun_out/gosab/$lib$rep.json').read().strip().splitlines()[-1])
print('$lib rep $rep value', r['value'], 'launch_ms', r['roofline']['avg_launch_ms'], 'chip_frac', r['roofline']['chip_frac'])"
done; done

#!/bin/bash
# Go-19 40x256 f16 B=512 through the product library ("new") and alternative builds kzero_amd/libkzhip_<name>.so,
# alternating, plus the Go / board-conv parity tests.  Usage (GPU box): LIBS="base new nt" bash tools/go_ab.sh
mkdir -p gpurun_out/goab
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "go or board_conv or g8 or per_layer" 2>&1 | tail -2
for rep in 1 2; do for lib in ${LIBS:-base new}; do
  if [ $lib != new ]; then [ -f kzero_amd/libkzhip_$lib.so ] || continue; export KZ_LIB_PATH=$PWD/kzero_amd/libkzhip_$lib.so; else unset KZ_LIB_PATH; fi
  python bench.py --workload go19-40x256 --dtype f16 --steps ${STEPS:-150} --warmup 10 --no-cpu-baseline --no-others --no-seam --no-host-io > gpurun_out/goab/$lib$rep.json 2> gpurun_out/goab/$lib$rep.err
  python -c "
import json
r=json.loads(open('gpurun_out/goab/$lib$rep.json').read().strip().splitlines()[-1])
print('$lib rep $rep value', r['value'], 'launch_ms', r['roofline']['avg_launch_ms'], 'chip_frac', r['roofline']['chip_frac'])"
done; done

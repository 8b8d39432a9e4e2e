#!/bin/bash
# Chess 20x256 through KZ_DTYPE_F32_SPLIT16: the product library ("new") against alternative builds
# kzero_amd/libkzhip_<name>.so and against engine counts, alternating.  Usage (GPU box): LIBS="new wb0" ENGINES="1 2 3" bash tools/split_ab.sh
mkdir -p gpurun_out/splitab
for rep in 1 2; do for lib in ${LIBS:-new}; do for eng in ${ENGINES:-2}; do
  if [ $lib != new ]; then [ -f kzero_amd/libkzhip_$lib.so ] || continue; export KZ_LIB_PATH=$PWD/kzero_amd/libkzhip_$lib.so; else unset KZ_LIB_PATH; fi
  python bench.py --workload chess-20x256 --dtype f32split16 --engines $eng --steps ${STEPS:-1500} --warmup 20 --no-cpu-baseline --no-others --no-seam ${EXTRA:-} > gpurun_out/splitab/$lib.$eng.$rep.json 2> gpurun_out/splitab/$lib.$eng.$rep.err
  python -c "
import json
r=json.loads(open('gpurun_out/splitab/$lib.$eng.$rep.json').read().strip().splitlines()[-1])
print('$lib engines $eng rep $rep value', r['value'], 'launch_ms', r['roofline']['avg_launch_ms'], 'pcie', (r.get('pcie_inclusive') or {}).get('value'))"
done; done; done

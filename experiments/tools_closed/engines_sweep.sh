#!/bin/bash
# Chess 20x256 f16 B=256 device-resident against the number of engines (streams) per GPU.  Usage: bash tools/engines_sweep.sh
mkdir -p gpurun_out/eng
for rep in 1 2; do for e in 2 3 4; do
  python bench.py --engines $e --steps 3000 --warmup 50 --no-cpu-baseline --no-others --no-seam --no-host-io > gpurun_out/eng/e$e$rep.json 2> gpurun_out/eng/e$e$rep.err
  python -c "
import json
r=json.loads(open('gpurun_out/eng/e$e$rep.json').read().strip().splitlines()[-1])
print('engines $e rep $rep value', r['value'], 'launch_ms', r['roofline']['avg_launch_ms'], 'chip_frac', r['roofline']['chip_frac'])"
done; done

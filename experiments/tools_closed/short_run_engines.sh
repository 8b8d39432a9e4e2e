#!/bin/bash
# The driver's short run (--steps 20 --warmup 5) with 2 and 4 engines, five times each, alternating.
mkdir -p gpurun_out/short
for rep in 1 2 3 4 5; do for e in 2 4; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --engines $e --no-cpu-baseline --no-others --no-seam > gpurun_out/short/e$e$rep.json 2> gpurun_out/short/e$e$rep.err
  python -c "
import json
r=json.loads(open('gpurun_out/short/e$e$rep.json').read().strip().splitlines()[-1])
print('engines $e rep $rep value', r['value'], 'ms_per_step', r['ms_per_step'], 'pcie', r['pcie_inclusive']['value'])"
done; done

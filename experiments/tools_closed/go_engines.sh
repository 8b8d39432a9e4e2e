#!/bin/bash
# Go-19 40x256 f16 B=512 against the number of engines (streams).  Usage (GPU box): bash tools/go_engines.sh
mkdir -p gpurun_out/goeng
for rep in 1 2; do for e in 1 2 3; do
  python bench.py --workload go19-40x256 --dtype f16 --engines $e --steps 150 --warmup 10 --no-cpu-baseline --no-others --no-seam --no-host-io > gpurun_out/goeng/e$e$rep.json 2> gpurun_out/goeng/e$e$rep.err
  python -c "
import json
r=json.loads(open('gpurun_out/goeng/e$e$rep.json').read().strip().splitlines()[-1])
print('engines $e rep $rep value', r['value'], 'launch_ms', r['roofline']['avg_launch_ms'], 'chip_frac', r['roofline']['chip_frac'])"
done; done

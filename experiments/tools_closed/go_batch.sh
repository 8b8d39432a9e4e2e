#!/bin/bash
# Go-19 40x256 f16 against the batch size (do the per-layer activations fit the 256 MB Infinity Cache?)
mkdir -p gpurun_out/gobatch
for rep in 1 2; do for b in 128 192 256 384 512; do
  python bench.py --workload go19-40x256 --dtype f16 --batch $b --steps 100 --warmup 10 --no-cpu-baseline --no-others --no-seam --no-host-io > gpurun_out/gobatch/b$b$rep.json 2> gpurun_out/gobatch/b$b$rep.err
  python -c "
import json
r=json.loads(open('gpurun_out/gobatch/b$b$rep.json').read().strip().splitlines()[-1])
print('batch $b rep $rep value', r['value'], 'launch_ms', r['roofline']['avg_launch_ms'], 'us/board/layer', 1e3*r['roofline']['avg_launch_ms']/$b)"
done; done

#!/bin/bash
# (experiment kernels: needs the experiment build, experiments/build.sh)
export KZ_LIB_PATH=${KZ_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/experiments/libkzhip_exp.so}
# hipGraph replay (KZ_HIP_GRAPH=1) against eager launches on the multi-launch paths: Go-19 40x256 f16 B=512 (85 launches
# per batch), Ataxx 8x128 f16 B=256 (one-launch tower + 4 head launches, host-bound), device-resident and PCIe-inclusive.
# Usage (GPU box): bash tools/graph_ab.sh
mkdir -p gpurun_out/graph
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hip_graph" 2>&1 | tail -2
for g in 0 1; do
  KZ_BENCH_NO_KERNEL_TIMING=1 KZ_HIP_GRAPH=$g python bench.py --workload go19-40x256 --dtype f16 --steps 150 --warmup 10 --no-cpu-baseline --no-others --no-seam > gpurun_out/graph/go_g$g.json 2> gpurun_out/graph/go_g$g.err
  KZ_BENCH_NO_KERNEL_TIMING=1 KZ_HIP_GRAPH=$g python bench.py --workload ataxx-8x128 --dtype f16 --steps 6000 --warmup 100 --no-cpu-baseline --no-others --no-seam > gpurun_out/graph/a1f16_g$g.json 2> gpurun_out/graph/a1f16_g$g.err
  for w in go a1f16; do python -c "
import json
r=json.loads(open('gpurun_out/graph/${w}_g$g.json').read().strip().splitlines()[-1])
print('$w graph=$g value', r['value'], 'pcie_inclusive', r.get('pcie_inclusive',{}).get('value'), 'engines', r['config'].get('engines_per_gpu'))"; done
done

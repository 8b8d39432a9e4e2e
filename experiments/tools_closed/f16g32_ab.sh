#!/bin/bash
# (experiment kernels: needs the experiment build, experiments/build.sh)
export KZ_LIB_PATH=${KZ_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/experiments/libkzhip_exp.so}
# Chess 20x256 plain f16 through the generic one-launch tower (KZ_NO_TOWER_F16=1: tower_resident_f16g, one board per
# workgroup, separate head launches): 16x16x32 against 32x32x16 MFMAs — does the lower register-file traffic per FLOP
# of the large tile raise the power-limited rate in a real kernel?  Usage (GPU box): bash tools/f16g32_ab.sh
mkdir -p gpurun_out/g32
export KZ_NO_TOWER_F16=1
for rep in 1 2; do for m32 in 0 1; do
  KZ_F16G_MFMA32=$m32 python bench.py --workload chess-20x256 --dtype f16 --steps 600 --warmup 20 --no-cpu-baseline --no-others --no-seam --no-host-io > gpurun_out/g32/m$m32$rep.json 2> gpurun_out/g32/m$m32$rep.err
  python -c "
import json
r=json.loads(open('gpurun_out/g32/m$m32$rep.json').read().strip().splitlines()[-1])
print('mfma32=$m32 rep $rep value', r['value'], 'launch_ms', r['roofline']['avg_launch_ms'], r['config'].get('tower_path'))"
done; done

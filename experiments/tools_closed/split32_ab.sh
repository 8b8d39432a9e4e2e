#!/bin/bash
# (experiment kernels: needs the experiment build, experiments/build.sh)
export KZ_LIB_PATH=${KZ_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/experiments/libkzhip_exp.so}
# Chess 20x256 through KZ_DTYPE_F32_SPLIT16: the 16x16x32 launch (default) against the 32x32x16 one (KZ_SPLIT_MFMA32=1),
# alternating, plus the split parity tests on both.  Usage (GPU box): bash tools/split32_ab.sh
mkdir -p gpurun_out/s32
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split16 or c1_f32 or range_overflow" 2>&1 | tail -2
KZ_SPLIT_MFMA32=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split16" 2>&1 | tail -1
for rep in 1 2; do for m16 in 0 1; do
  KZ_SPLIT_MFMA32=$m16 python bench.py --workload chess-20x256 --dtype f32split16 --steps 400 --warmup 20 --no-cpu-baseline --no-others --no-seam --no-host-io > gpurun_out/s32/m$m16$rep.json 2> gpurun_out/s32/m$m16$rep.err
  python -c "
import json
r=json.loads(open('gpurun_out/s32/m$m16$rep.json').read().strip().splitlines()[-1])
print('mfma32=$m16 rep $rep value', r['value'], 'launch_ms', r['roofline']['avg_launch_ms'])"
done; done

#!/bin/bash
# (experiment kernels: needs the experiment build, experiments/build.sh)
export KZ_LIB_PATH=${KZ_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/experiments/libkzhip_exp.so}
# NB=4 tower-only, one launch filling the chip (batch 1024 = 256 workgroups): product vs ablation builds, same box
export KZ_NO_FUSED_HEADS=1 KZ_TOWER_NB=4
for r in 1 2; do for v in "" _NOXLOAD _NOXIO; do
KZ_LIB_PATH=/root/repo/kzero_amd/libkzhip$v.so python bench.py --no-others --no-cpu-baseline --no-host-io --steps 800 --warmup 30 --engines 1 --batch 1024 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('lib$v evals/s', r['value'], 'launch ms', r['roofline']['avg_launch_ms'])"
done; done

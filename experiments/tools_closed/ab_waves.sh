#!/bin/bash
# (experiment kernels: needs the experiment build, experiments/build.sh)
export KZ_LIB_PATH=${KZ_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/experiments/libkzhip_exp.so}
# A/B of the tower-only launch (KZ_NO_FUSED_HEADS=1) with 4 and 8 waves per workgroup: identical outputs, then timing
# interleaved on one box.
export KZ_NO_FUSED_HEADS=1
cat > /tmp/dump.py <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from kzero_amd import capi, synth
blob = synth.random_model("chess", 3, 256, "attention", seed=81)
bits, sc = synth.random_boards("chess", 37, seed=82)
e = capi.Engine(capi.Model(blob=blob), 0, 64, capi.KZ_DTYPE_F16)
s, p = e.eval_packed(bits, sc)
np.save(sys.argv[1], np.concatenate([s.ravel(), p.ravel()]))
print(e.tower_path, float(np.abs(p).max()))
PY
KZ_TOWER_WAVES=4 python /tmp/dump.py /tmp/w4.npy
KZ_TOWER_WAVES=8 python /tmp/dump.py /tmp/w8.npy
python -c "
import numpy as np
a,b=np.load('/tmp/w4.npy'),np.load('/tmp/w8.npy'); print('max |w8-w4| =', np.abs(a-b).max(), 'finite', np.isfinite(b).all())"
for rep in 1 2; do for w in 4 8; do
KZ_TOWER_WAVES=$w python bench.py --no-others --no-cpu-baseline --no-host-io --steps 3000 --warmup 50 ${BENCH_ARGS:-} | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('waves $w', r['config']['tower_path'], 'evals/s', r['value'], 'launch ms', r['roofline']['avg_launch_ms'])"
done; done

#!/bin/bash
# (experiment kernels: needs the experiment build, experiments/build.sh)
export KZ_LIB_PATH=${KZ_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/experiments/libkzhip_exp.so}
# A/B of the tower-only launch (KZ_NO_FUSED_HEADS=1) with 2 and 4 boards per workgroup: identical outputs, then timing
# interleaved on one box.   tools/ab_nb4.sh [engines for NB=4, default 4]
export KZ_NO_FUSED_HEADS=1
E4=${1:-4}
cat > /tmp/dump.py <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from kzero_amd import capi, synth
out = []
for depth, n in ((1, 3), (3, 37), (20, 256)):
    blob = synth.random_model("chess", depth, 256, "attention", seed=81)
    bits, sc = synth.random_boards("chess", n, seed=82)
    e = capi.Engine(capi.Model(blob=blob), 0, 256, capi.KZ_DTYPE_F16)
    s, p = e.eval_packed(bits, sc)
    out += [s.ravel(), p.ravel()]
    print(e.tower_path, e.launch_geometry(n), depth, n, float(np.abs(p).max()), bool(np.isfinite(p).all()))
np.save(sys.argv[1], np.concatenate(out))
PY
KZ_TOWER_NB=2 python /tmp/dump.py /tmp/n2.npy
KZ_TOWER_NB=4 python /tmp/dump.py /tmp/n4.npy
python -c "
import numpy as np
a,b=np.load('/tmp/n2.npy'),np.load('/tmp/n4.npy'); print('max |nb4-nb2| =', np.abs(a-b).max(), 'n', a.size, 'finite', np.isfinite(b).all())"
for rep in 1 2; do
KZ_TOWER_NB=2 python bench.py --no-others --no-cpu-baseline --no-host-io --steps 3000 --warmup 50 --engines 2 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('NB=2 x2 engines', r['config']['tower_path'], 'evals/s', r['value'], 'launch ms', r['roofline']['avg_launch_ms'], 'wgs', r['roofline']['workgroups_per_launch'])"
KZ_TOWER_NB=4 python bench.py --no-others --no-cpu-baseline --no-host-io --steps 3000 --warmup 50 --engines $E4 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('NB=4 x$E4 engines', r['config']['tower_path'], 'evals/s', r['value'], 'launch ms', r['roofline']['avg_launch_ms'], 'wgs', r['roofline']['workgroups_per_launch'])"
done

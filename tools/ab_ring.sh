for v in "" _pre "" _pre; do echo "lib$v"; KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so python tools/shape_sweep.py --no-oracle --dtypes f16,parity --engines 2,3 --filter "$1" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print('  ', r['case'][:40], r['arith'], r.get('rate_path'), round(r.get('evals_per_s',0)), r.get('frac_of_peak'), r.get('engines'))
"; done

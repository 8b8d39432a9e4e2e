#!/bin/bash
# (experiment kernels: needs the experiment build, experiments/build.sh)
export KZ_LIB_PATH=${KZ_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/experiments/libkzhip_exp.so}
# clock / MFMA-busy of the split launch on chess 20x256 (batch 256 = 256 workgroups: a full chip), 32x32x16 and 16x16x32
export TMPDIR=/tmp
for m16 in 0 1; do
out=$PWD/gpurun_out/pmc_split_m$m16
rm -rf $out
KZ_SPLIT_MFMA32=$m16 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $out -o run -- python3 bench.py --workload chess-20x256 --dtype f32split16 --no-cpu-baseline --no-others --no-host-io --no-seam --engines 1 --steps 100 --warmup 10 > $out.log 2>&1
python3 - "$out/run_counter_collection.csv" $m16 <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if 'tower_resident_split' in r['Kernel_Name']:
        agg[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
def mean(name):
    v=agg[name][len(agg[name])//2:]
    return sum(x for x,_ in v)/len(v), sum(t for _,t in v)/len(v)
g,t=mean('GRBM_GUI_ACTIVE'); m,_=mean('SQ_VALU_MFMA_BUSY_CYCLES')
cyc=g/8
print('split mfma32=%s launch_us %.1f clock_GHz %.3f cycles %.0f mfma_busy %.3f busy_GHz %.3f' % (sys.argv[2], t/1e3, cyc/t, cyc, m/(cyc*1024), m/(cyc*1024)*cyc/t))
PY
done

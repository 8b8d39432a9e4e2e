#!/bin/bash
# clock / MFMA-busy of the exact-f32 resident launch at A1 (Ataxx 8x128, batch 512 = 256 workgroups: a full chip)
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_a1_clk
rm -rf $out
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $out -o run -- python3 bench.py --workload ataxx-8x128 --dtype f32 --no-cpu-baseline --no-others --no-host-io --batch ${BATCH:-512} --engines 1 --steps 200 --warmup 20 > $out.log 2>&1
python3 - "$out/run_counter_collection.csv" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if 'tower_resident_f32' in r['Kernel_Name']:
        agg[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
def mean(name):
    v=agg[name][len(agg[name])//2:]
    return sum(x for x,_ in v)/len(v), sum(t for _,t in v)/len(v)
g,t=mean('GRBM_GUI_ACTIVE'); m,_=mean('SQ_VALU_MFMA_BUSY_CYCLES')
cyc=g/8
print('A1 f32 launch_us %.1f clock_GHz %.3f cycles %.0f mfma_busy %.3f' % (t/1e3, cyc/t, cyc, m/(cyc*1024)))
PY

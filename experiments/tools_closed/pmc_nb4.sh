#!/bin/bash
# (experiment kernels: needs the experiment build, experiments/build.sh)
export KZ_LIB_PATH=${KZ_LIB_PATH:-$(cd "$(dirname "$0")/../.." && pwd)/experiments/libkzhip_exp.so}
# in-load clock and MFMA-busy share of the tower-only launch at a FULL chip: NB=2 (batch 512) vs NB=4 (batch 1024)
export TMPDIR=/tmp KZ_NO_FUSED_HEADS=1
for cfg in "2 512" "4 1024"; do set -- $cfg; nb=$1; batch=$2
 for set in clk sq; do
  case $set in
    clk) ctr="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES";;
    sq) ctr="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY";;
  esac
  out=$PWD/gpurun_out/pmc_nb${nb}_$set
  rm -rf $out
  KZ_TOWER_NB=$nb rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python3 bench.py --no-cpu-baseline --no-others --no-host-io --batch $batch --engines 1 --steps 200 --warmup 20 > $out.log 2>&1
  python3 - "$out/run_counter_collection.csv" "NB=$nb batch $batch" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if 'tower_resident' in r['Kernel_Name']:
        agg[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
def mean(name):
    v=agg[name][len(agg[name])//2:]
    return sum(x for x,_ in v)/len(v), sum(t for _,t in v)/len(v)
if 'GRBM_GUI_ACTIVE' in agg:
    g,t=mean('GRBM_GUI_ACTIVE'); m,_=mean('SQ_VALU_MFMA_BUSY_CYCLES')
    cyc=g/8
    print(sys.argv[2], 'launch_us %.1f clock_GHz %.3f cycles %.0f mfma_busy %.3f' % (t/1e3, cyc/t, cyc, m/(cyc*1024)))
else:
    print(sys.argv[2], ' '.join('%s %.4g' % (k, mean(k)[0]) for k in sorted(agg)), 'launch_us %.1f' % (mean('SQ_WAVE_CYCLES')[1]/1e3))
PY
 done
done

#!/bin/bash
# Counter passes of a workload's dominant kernel (one engine; counter collection serialises launches):
#   tools/pmc_workload.sh <workload> <dtype> [steps]      results under gpurun_out/pmcw_<workload>_<dtype>_<set>/
# sets: clk (in-load clock = GRBM_GUI_ACTIVE / 8 / duration, MFMA-busy share), sq (issue / wait / LDS), l2 (TCC hit / miss)
export TMPDIR=/tmp
WL=$1; DT=$2; STEPS=${3:-30}
for set in clk sq l2; do
  case $set in
    clk) ctr="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES";;
    sq) ctr="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY";;
    l2) ctr="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum";;
  esac
  out=$PWD/gpurun_out/pmcw_${WL}_${DT}_$set
  rm -rf $out
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python3 bench.py --repeats 1 --workload $WL --dtype $DT --no-cpu-baseline --no-others --boundary resident --no-host-io --no-seam --engines 1 --steps $STEPS --warmup 5 --prewarm 0 > $out.log 2>&1
  python3 - "$out/run_counter_collection.csv" "$WL $DT $set" <<'PY'
import csv,sys,collections
try:
    rows=list(csv.DictReader(open(sys.argv[1])))
except OSError as e:
    print(sys.argv[2], "no counters:", e); sys.exit(0)
agg=collections.defaultdict(list)
for r in rows:
    if 'tower' in r['Kernel_Name'] or 'board_conv' in r['Kernel_Name']:
        agg[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
def mean(name):
    v=agg[name][len(agg[name])//2:]
    return sum(x for x,_ in v)/len(v), sum(t for _,t in v)/len(v)
if 'GRBM_GUI_ACTIVE' in agg:
    g,t=mean('GRBM_GUI_ACTIVE'); m,_=mean('SQ_VALU_MFMA_BUSY_CYCLES')
    cyc=g/8
    print(sys.argv[2], 'launch_us %.1f clock_GHz %.3f cycles %.0f mfma_busy %.3f' % (t/1e3, cyc/t, cyc, m/(cyc*1024)))
else:
    print(sys.argv[2], ' '.join('%s %.4g' % (k, mean(k)[0]) for k in sorted(agg)), 'launch_us %.1f' % (mean(sorted(agg)[0])[1]/1e3 if agg else 0))
PY
done

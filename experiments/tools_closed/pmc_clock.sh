#!/bin/bash
# PMC passes of the tower launch at a full chip (batch 512 = 256 workgroups, one engine; counters serialise launches,
# so two concurrent half-chip launches cannot be measured this way):
#   tools/pmc_clock.sh <lib suffix...>     results under gpurun_out/pmc_<set><suffix>/
# set "clk": in-load clock = GRBM_GUI_ACTIVE / 8 / duration, MFMA-busy share of SIMD cycles; set "sq": LDS and issue.
export TMPDIR=/tmp
SETS=${SETS:-clk sq}
for v in "$@"; do
 for set in $SETS; do
  case $set in
    clk) ctr="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES";;
    sq) ctr="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY";;
  esac
  out=$PWD/gpurun_out/pmc_$set$v
  rm -rf $out
  KZ_LIB_PATH=/root/repo/kzero_amd/libkzhip$v.so rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python3 bench.py --repeats 1 --no-cpu-baseline --batch ${BATCH:-512} --engines 1 --steps 300 --warmup 20 ${BENCH_ARGS:-} > $out.log 2>&1
  python3 - "$out/run_counter_collection.csv" "lib$v" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
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
    print(sys.argv[2], ' '.join('%s %.4g' % (k, mean(k)[0]) for k in sorted(agg)), 'launch_us %.1f' % (mean('SQ_WAVE_CYCLES')[1]/1e3))
PY
 done
done

#!/bin/bash
# clock / MFMA-busy of kz_board_conv_f16 on Go-19 40x256 B=512 (one engine)
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_go_clk
rm -rf $out
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $out -o run -- python3 bench.py --repeats 1 --workload go19-40x256 --dtype f16 --no-cpu-baseline --no-others --boundary resident --no-host-io --no-seam --engines 1 --steps 4 --warmup 1 --prewarm 0 > $out.log 2>&1
python3 - "$out/run_counter_collection.csv" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if 'kz_board_conv' in r['Kernel_Name']:
        agg[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
def mean(name):
    v=agg[name][len(agg[name])//2:]
    return sum(x for x,_ in v)/len(v), sum(t for _,t in v)/len(v)
g,t=mean('GRBM_GUI_ACTIVE'); m,_=mean('SQ_VALU_MFMA_BUSY_CYCLES')
cyc=g/8
print('Go board_conv launch_us %.1f clock_GHz %.3f cycles %.0f mfma_busy %.3f busy_GHz %.3f' % (t/1e3, cyc/t, cyc, m/(cyc*1024), m/(cyc*1024)*cyc/t))
PY

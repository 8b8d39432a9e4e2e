#!/bin/bash
# The Winograd F(2x2,3x3) inner-loop probe (tools/micro/winograd_probe.hip) on a full chip: microseconds per four-board
# layer by HIP events, then clock / MFMA-busy / LDS-busy from a separate rocprofv3 --pmc pass (never combined with traces).
export TMPDIR=/tmp
mkdir -p gpurun_out/winograd
[ -x tools/micro/winograd_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/micro/winograd_probe tools/micro/winograd_probe.hip
./tools/micro/winograd_probe | tee gpurun_out/winograd/probe.txt
out=$PWD/gpurun_out/winograd/pmc
rm -rf $out
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $out -o run -- ./tools/micro/winograd_probe > $out.log 2>&1
python3 - "$out/run_counter_collection.csv" <<'PY' | tee gpurun_out/winograd/pmc.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r['Kernel_Name']][r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
for k,c in agg.items():
    def mean(name):
        v=c[name][len(c[name])//2:]
        return sum(x for x,_ in v)/max(len(v),1), sum(t for _,t in v)/max(len(v),1)
    g,t=mean('GRBM_GUI_ACTIVE'); m,_=mean('SQ_VALU_MFMA_BUSY_CYCLES'); l,_=mean('SQ_ACTIVE_INST_LDS')
    cyc=g/8
    print('%s: launch_us %.1f clock_GHz %.3f mfma_busy %.3f lds_active_share %.3f' % (k[:60], t/1e3, cyc/t, m/(cyc*1024) if cyc else 0, l/(cyc*1024) if cyc else 0))
PY

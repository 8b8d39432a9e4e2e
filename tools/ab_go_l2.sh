#!/bin/bash
# Round 6, VERDICT r5 #4: what is the activation traffic's share of the Go launch's time / power?
# Same-box A/B on Go 19x19 40x256 b=512 f16, device-resident, interleaved:
#   libkzhip_prev.so  round 5's kernel (tools/build_rev_lib.sh <rev> prev)
#   libkzhip.so       the working tree's (conflict-free staging stores)
#   libkzhip_l2.so    -DKZ_BC_L2_ABLATE: every workgroup of an XCD stages from / stores to ONE board (served by that XCD's
#                     L2; same instruction stream, wrong results) — the kernel with its HBM traffic taken away
# then the clock / MFMA-busy / LDS counters of each build (one engine, serialised launches).
export TMPDIR=/tmp
O=gpurun_out/r6_go_l2; mkdir -p $O
for r in 1 2 3; do for v in _prev "" _l2; do
  KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so python3 bench.py --workload go19-40x256 --dtype f16 --boundary resident --no-cpu-baseline --no-host-io --no-others --no-seam --repeats 3 --steps ${STEPS:-200} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"go19-40x256 f16 lib$v\", d[\"value\"], d[\"value_min\"], d[\"value_max\"], d[\"roofline\"][\"avg_launch_ms\"])"
done; done | tee $O/ab.txt
for v in _prev "" _l2; do
  for set in clk sq; do
    case $set in
      clk) ctr="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES";;
      sq) ctr="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY";;
    esac
    out=$PWD/$O/pmc${v}_$set; rm -rf $out
    KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out -o run -- python3 bench.py --repeats 1 --workload go19-40x256 --dtype f16 --boundary resident --no-cpu-baseline --no-others --no-host-io --no-seam --engines 1 --steps 3 --warmup 1 --prewarm 0 > $out.log 2>&1
    python3 - "$out/run_counter_collection.csv" "lib$v $set" <<'PY'
import csv,sys,collections
try:
    rows=list(csv.DictReader(open(sys.argv[1])))
except OSError as e:
    print(sys.argv[2], "no counters:", e); sys.exit(0)
agg=collections.defaultdict(list)
for r in rows:
    if 'board_conv' in r['Kernel_Name']:
        agg[r['Counter_Name']].append((float(r['Counter_Value']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
def mean(name):
    v=agg[name][len(agg[name])//2:]
    return sum(x for x,_ in v)/len(v), sum(t for _,t in v)/len(v)
if 'GRBM_GUI_ACTIVE' in agg:
    g,t=mean('GRBM_GUI_ACTIVE'); m,_=mean('SQ_VALU_MFMA_BUSY_CYCLES')
    cyc=g/8
    print(sys.argv[2], 'launch_us %.1f clock_GHz %.3f cycles %.0f mfma_busy %.3f busy_GHz %.3f' % (t/1e3, cyc/t, cyc, m/(cyc*1024), m/(cyc*1024)*cyc/t))
else:
    print(sys.argv[2], ' '.join('%s %.4g' % (k, mean(k)[0]) for k in sorted(agg)), 'launch_us %.1f' % (mean(sorted(agg)[0])[1]/1e3 if agg else 0))
PY
    rm -rf $out
  done
done | tee $O/counters.txt

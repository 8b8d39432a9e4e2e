#!/bin/bash
# LDS counters of kz_board_conv_f16 on Go-19 40x256 B=512 per library build (one engine, serialised launches), residual and
# non-residual layers separately:   tools/pmc_go_lds.sh "" _d1 _d2 ...   (suffixes of kzero_amd/libkzhip<suffix>.so;
# _dN = -DKZ_BC_DIAG=N phase-skip builds: which phase the remaining SQ_LDS_BANK_CONFLICT cycles belong to)
export TMPDIR=/tmp
for v in "$@"; do
  out=$PWD/gpurun_out/pmc_go_lds$v; rm -rf $out
  KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $out -o run -- python3 bench.py --repeats 1 --workload go19-40x256 --dtype f16 --boundary resident --no-cpu-baseline --no-others --no-host-io --no-seam --engines 1 --steps 2 --warmup 1 --prewarm 0 > $out.log 2>&1
  python3 - "$out/run_counter_collection.csv" "lib$v" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'board_conv_f16' in r['Kernel_Name']]
# dispatches in order: per batch 81 launches: stem, then (conv A, conv B with residual) x 40
by=collections.defaultdict(dict)
for r in rows: by[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(by)
last=ids[-81:]
def mean(sel,name):
    v=[by[i][name] for i in sel if name in by[i]]
    return sum(v)/max(len(v),1)
a=[last[k] for k in range(1,81,2)]; b=[last[k] for k in range(2,81,2)]
for lab,sel in (("convA (no residual)",a),("convB (residual)",b)):
    print(sys.argv[2], lab, ' '.join('%s %.4g' % (n, mean(sel,n)) for n in ('SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','SQ_INSTS_LDS')))
PY
  rm -rf $out
done

#!/bin/bash
# HBM-side bytes per launch of a workload's dominant kernel: two separate rocprofv3 --pmc passes (FETCH_SIZE, then
# WRITE_SIZE; --pmc with --kernel-trace only, never with the sys / hip / hsa trace domains), summarised into profiles/hbm_traffic.json by tools/pmc_traffic.py.
#   [STEPS=60 WARMUP=10] tools/pmc_traffic.sh chess-20x256 f16 [batch]      (per-layer paths: STEPS=4 WARMUP=1 — a Go
#   batch is 85 launches and counter collection serialises every one of them)
# One engine (counter collection serialises launches); run on the GPU box from the repo root.
set -u
export TMPDIR=/tmp
WL=${1:-chess-20x256}; DT=${2:-f16}; BATCH=${3:-}
TAG=${WL}_${DT}
for c in FETCH_SIZE WRITE_SIZE; do
  out=$PWD/gpurun_out/pmc_traffic_${TAG}_$c
  rm -rf "$out"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out" -o run -- python3 bench.py --repeats 1 --workload $WL --dtype $DT \
     ${BATCH:+--batch $BATCH} --engines 1 --steps ${STEPS:-60} --warmup ${WARMUP:-10} --prewarm 0 --boundary resident --no-cpu-baseline --no-others --no-host-io \
     > "$out.log" 2>&1
done
python3 tools/pmc_traffic.py "$WL" "$DT" "${BATCH:-0}" "$PWD/gpurun_out/pmc_traffic_${TAG}_FETCH_SIZE" "$PWD/gpurun_out/pmc_traffic_${TAG}_WRITE_SIZE"

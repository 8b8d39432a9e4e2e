#!/bin/bash
# Cost of the fused heads inside the exact-f32 launch: one engine, batch 512 (a full chip per launch), the tower kernel's
# average launch with and without the heads.  Usage (GPU box): bash tools/a1_tail_cost.sh
mkdir -p gpurun_out/a1t
for nf in 0 1; do
  KZ_NO_FUSED_HEADS=$nf python bench.py --workload ataxx-8x128 --dtype f32 --engines 1 --batch 512 --steps 600 --warmup 30 \
     --no-cpu-baseline --no-others --no-seam --no-host-io > gpurun_out/a1t/nf$nf.json 2> gpurun_out/a1t/nf$nf.err
  python - <<PY
import json
r=json.loads(open("gpurun_out/a1t/nf$nf.json").read().strip().splitlines()[-1])
print("nofuse=$nf value", r["value"], "launch_ms", r["roofline"]["avg_launch_ms"], "path", r["config"].get("tower_path"))
PY
done

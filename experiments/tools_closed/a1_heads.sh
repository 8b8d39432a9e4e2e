#!/bin/bash
# A1 (Ataxx 8x128 exact f32, batch 256) with the heads inside the tower launch against separate head launches, 2 and 3
# engines; then the f32 parity tests.  Usage (GPU box): bash tools/a1_heads.sh
mkdir -p gpurun_out/a1h
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "f32 or a1 or golden or split16 or range" 2>&1 | tail -3
for nf in 0 1; do for e in 2 3; do
  KZ_NO_FUSED_HEADS=$nf python bench.py --workload ataxx-8x128 --dtype f32 --engines $e --steps 1500 --warmup 50 \
     --no-cpu-baseline --no-others --no-seam > gpurun_out/a1h/a1_nf${nf}_e$e.json 2> gpurun_out/a1h/a1_nf${nf}_e$e.err
  python - <<PY
import json
r=json.loads(open("gpurun_out/a1h/a1_nf${nf}_e$e.json").read().strip().splitlines()[-1])
print("nofuse=$nf engines=$e value", r["value"], "pcie", r.get("pcie_inclusive"), "path", r["config"].get("tower_path"), "chip_frac", r["roofline"].get("chip_frac"), "frac", r["roofline"]["frac"])
PY
done; done
python bench.py --workload chess-20x256 --dtype f32 --steps 300 --warmup 20 --no-cpu-baseline --no-others --no-seam | python tools/show_bench.py 2>/dev/null | head -5

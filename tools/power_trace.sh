#!/bin/bash
# Socket power and shader clock (rocm-smi, sampled every 0.5 s) while each workload runs device-resident for ~8 s:
# the direct reading behind the busy-GHz argument of DESIGN.md §5.1 (cap: 1400 W).
mkdir -p gpurun_out/power
run() {  # name, bench args...
  name=$1; shift
  python bench.py "$@" --no-cpu-baseline --no-others --no-seam --boundary resident --no-host-io > gpurun_out/power/$name.json 2> gpurun_out/power/$name.err &
  pid=$!
  sleep 4
  for i in 1 2 3 4 5 6; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ' '; echo
    sleep 0.5
  done > gpurun_out/power/$name.smi
  wait $pid
  python3 - "$name" <<'PY'
import re, sys, json
name = sys.argv[1]
p, c = [], []
for ln in open(f"gpurun_out/power/{name}.smi"):
    m = re.search(r"Power \(W\): ([0-9.]+)", ln); n = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", ln)
    if m: p.append(float(m.group(1)))
    if n: c.append(int(n.group(1)))
r = json.loads(open(f"gpurun_out/power/{name}.json").read().strip().splitlines()[-1])
print(f"{name:28s} value {r['value']:10.1f}  chip_frac {r['roofline']['chip_frac']:.3f}  power W mean {sum(p)/max(len(p),1):7.1f} max {max(p or [0]):7.1f}  sclk MHz {sorted(c)}")
PY
}
run chess_f16        --workload chess-20x256 --dtype f16 --steps 24000 --warmup 50
run chess_f32split16 --workload chess-20x256 --dtype f32split16 --steps 7000 --warmup 20
run chess_f32        --workload chess-20x256 --dtype f32 --steps 2200 --warmup 10
run ataxx_f32        --workload ataxx-8x128 --dtype f32 --steps 22000 --warmup 50
run go19_f16         --workload go19-40x256 --dtype f16 --steps 700 --warmup 5

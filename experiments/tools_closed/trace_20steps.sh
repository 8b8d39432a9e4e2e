#!/bin/bash
# Kernel trace of the driver's short run (--steps 20 --warmup 5): start / duration of the timed region's 20 tower launches.
export TMPDIR=/tmp
out=$PWD/gpurun_out/trace20
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o run -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-others --no-seam --no-cpu-baseline --no-host-io > $out/bench.json 2> $out/bench.err
python3 - "$out" <<'PY'
import csv, sys, glob, json
f = glob.glob(sys.argv[1] + "/**/run_kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "kz_tower_resident" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-20:]
t0 = int(last[0]["Start_Timestamp"])
prev_end = {}
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else float("nan")
    print("queue %s start %8.1f us dur %7.1f us gap-on-queue %6.1f us" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap))
    prev_end[q] = e
print("span of the 20 launches: %.1f us" % ((max(int(r["End_Timestamp"]) for r in last) - t0) / 1e3))
# the launches just before (conditioning + warm-up) for comparison
before = rows[-60:-20]
print("mean duration of the 40 launches before: %.1f us" % (sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in before) / len(before) / 1e3))
print(json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])["ms_per_step"])
PY

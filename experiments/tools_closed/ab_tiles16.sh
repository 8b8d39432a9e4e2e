#!/bin/bash
# same-box A/B of library builds on the 128-channel one-launch f16 towers at batches that fill the chip with three / four
# boards per workgroup (round 5: the sixteen-tile level): tools/ab_tiles16.sh _prev ""
LIBS=("$@")
for r in 1 2 3; do for v in "${LIBS[@]}"; do for w in "go9-16x128 600" "chess-20x128 600 2048" "chess-20x128 600 1024"; do
  read wl steps batch <<< "$w"
  KZ_LIB_PATH=$PWD/kzero_amd/libkzhip$v.so python bench.py --workload $wl ${batch:+--batch $batch} --dtype f16 --no-cpu-baseline --no-host-io --no-others --no-seam --repeats 3 --steps $steps 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$wl f16 lib$v\", d[\"value\"], d[\"value_min\"], d[\"value_max\"], d[\"roofline\"][\"avg_launch_ms\"], d[\"roofline\"][\"workgroups_per_launch\"], d[\"roofline\"][\"boards_per_workgroup\"])"
done; done; done

#!/bin/bash
# A1 in exact f32 (BASELINE configs[1]) with two and with three boards per workgroup (experiment build, KZ_T32_BOARDS=3)
# over engines per GPU, same box: tools/ab_a1_f32.sh
for r in 1 2; do for b in 2 3; do for e in 2 3 4; do
  KZ_T32_BOARDS=$b KZ_LIB_PATH=$PWD/experiments/libkzhip_exp.so python bench.py --workload ataxx-8x128 --dtype f32 --engines $e --no-cpu-baseline --no-host-io --no-others --no-seam --steps 4000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('boards $b engines $e', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['workgroups_per_launch'])"
done; done; done

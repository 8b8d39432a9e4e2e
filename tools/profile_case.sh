#!/bin/bash
# per-kernel time of ONE sweep case at rate depth (tools/shape_sweep.py), one engine:
#   tools/profile_case.sh <case substring> [dtypes, default f16]      -> gpurun_out/case_<substring>/kernel_stats.csv
export TMPDIR=/tmp
C=$1; D=${2:-f16}; O=gpurun_out/case_$C
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 tools/shape_sweep.py --no-oracle --filter "$C" --dtypes $D --engines 1 --seconds 0.5 > $O/sweep.log 2>&1
f=$(ls -S $(find $O/prof -name "*kernel_stats.csv") | head -1)
cp "$f" $O/kernel_stats.csv; rm -rf $O/prof
grep evals_per_s $O/sweep.log | cut -c1-300
head -14 $O/kernel_stats.csv | cut -c1-200

export TMPDIR=/tmp
O=gpurun_out/r5
mkdir -p $O
bash tools/bench_executor_r5.sh 1 > /dev/null 2>&1
rm -rf $O/stats_seam
KZ_BENCH_CLEAN_EXIT=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_seam -o run -- tests/cpp/build/bench_executor /tmp/chess20x256.kzm 3 1 6 256 8 f16 3 1 0 real 1 > $O/stats_seam.json 2> $O/stats_seam.log
echo rc=$?
f=$(find $O/stats_seam -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $O/kernel_stats_seam_default.csv || echo "no kernel stats of the seam run" > $O/kernel_stats_seam_default.csv
rm -rf $O/stats_seam
cat $O/kernel_stats_seam_default.csv | cut -c1-200
tail -2 $O/stats_seam.json | cut -c1-300

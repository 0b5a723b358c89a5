#!/bin/bash
# Round-5 baseline evidence taken at the round-4 HEAD library (run through gpurun):
#   1. C5 (2^28 u64 + u32) under rocprofv3 --kernel-trace: per-kernel durations of the timed sorts
#   2. C3 bench line on this box
set -x
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05
mkdir -p $OUT
make -C oracle -s > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_c5 -- python3 $R/tools/sort_loop.py --log2 28 --key-bytes 8 --steps 10 --warmup 3 > $OUT/c5_loop_under_rocprof.txt 2> $OUT/prof_c5.err
cd $R
python tools/trace_summary.py $(find $OUT/prof_c5 -name "*kernel_trace.csv" | head -1) --sorts 10 > $OUT/c5_timed_region_from_trace_baseline.txt
find $OUT/prof_c5 -name "*kernel_stats.csv" -exec cp {} $OUT/c5_kernel_stats_baseline.csv \;
rm -rf $OUT/prof_c5
python tools/sort_loop.py --log2 28 --key-bytes 8 > $OUT/c5_loop_baseline.txt 2>&1
python bench.py --no-cpu-baseline > $OUT/bench_n1_baseline.json 2> $OUT/bench_n1_baseline.err
ls -la $OUT

#!/bin/bash
# round 5, run AM: the bench under rocprofv3 --kernel-trace --stats at the last library build: kernel statistics + the timed region of the trace
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05am
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify > $OUT/bench_n1_under_rocprof.json 2> $OUT/prof_bench.err
cd $R
find $OUT/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/bench_n1_kernel_stats.csv \;
python tools/trace_summary.py $(find $OUT/prof_bench -name "*kernel_trace.csv" | head -1) --bench 10 > $OUT/bench_n1_timed_region_from_trace.txt
python tools/trace_summary.py $(find $OUT/prof_bench -name "*kernel_trace.csv" | head -1) --sorts 10 >> $OUT/bench_n1_timed_region_from_trace.txt
rm -rf $OUT/prof_bench
cat $OUT/bench_n1_timed_region_from_trace.txt | cut -c1-160

#!/bin/bash
# Round 6: kernel trace of the bench's timed region and of one headline sort (run on the GPU box through gpurun).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify > $OUT/bench_n1_under_rocprof.json 2> $OUT/prof_bench.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_loop -- python3 $R/tools/sort_loop.py --log2 28 --steps 3 --warmup 2 > $OUT/sort_loop_under_rocprof.txt 2> $OUT/prof_loop.err
cd $R
find $OUT/prof_bench -name "*kernel_stats.csv" -exec cp {} $OUT/bench_n1_kernel_stats.csv \;
python tools/trace_summary.py $(find $OUT/prof_bench -name "*kernel_trace.csv" | head -1) --bench 10 > $OUT/bench_n1_timed_region_from_trace.txt
python tools/trace_summary.py $(find $OUT/prof_bench -name "*kernel_trace.csv" | head -1) --sorts 10 >> $OUT/bench_n1_timed_region_from_trace.txt
python tools/trace_last_sort.py $(find $OUT/prof_loop -name "*kernel_trace.csv" | head -1) > $OUT/last_sort_kernels_2p28.txt
rm -rf $OUT/prof_bench $OUT/prof_loop
cat $OUT/bench_n1_timed_region_from_trace.txt $OUT/last_sort_kernels_2p28.txt

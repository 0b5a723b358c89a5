#!/bin/bash
# round 5, run K: after the sample kernel went parallel and the bookkeeping launches got lighter: parity, smoke, bench + trace
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05k
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python -m pytest tests/test_gpu_lds_finish.py tests/test_gpu_segmented_sort.py -x -q -m gpu > $OUT/t_fin_seg.log 2>&1
python bench.py --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python tools/sort_loop.py --log2 28 --key-bytes 8 > $OUT/c5_loop.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify --no-alt > $OUT/bench_n1_under_rocprof.json 2> $OUT/prof_bench.err
cd $R
python tools/trace_summary.py $(find $OUT/prof_bench -name "*kernel_trace.csv" | head -1) --sorts 10 > $OUT/bench_n1_timed_region_from_trace.txt
rm -rf $OUT/prof_bench

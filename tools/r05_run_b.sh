#!/bin/bash
# round 5, run B: parity (segmented, dist), a rank's compute at world 1 / as rank of 4 / of 8, kernel trace of the force-dist run
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05b
mkdir -p $OUT
python -m pytest tests/test_gpu_segmented_sort.py -x -q -m gpu > $OUT/t_seg.log 2>&1
python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1 > $OUT/fd27.json 2> $OUT/fd27.err
python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1 --as-rank-of 8 > $OUT/fd27_r8.json 2> $OUT/fd27_r8.err
GLU_HIP_SEG_SPLIT_GEO=4 python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1 --as-rank-of 8 > $OUT/fd27_r8_geo4.json 2> $OUT/fd27_r8_geo4.err
python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1 --as-rank-of 4 > $OUT/fd27_r4.json 2> $OUT/fd27_r4.err
python bench.py --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dist -- python3 $R/bench.py --force-dist --log2-keys 27 --steps 10 --warmup 3 --pipeline-depth 1 --no-verify --no-transport-fallback --no-one-gpu --no-cpu-baseline --as-rank-of 8 > $OUT/fd27_r8_under_rocprof.json 2> $OUT/prof_dist.err
cd $R
find $OUT/prof_dist -name "*kernel_stats.csv" -exec cp {} $OUT/force_dist_r8_kernel_stats.csv \;
find $OUT/prof_dist -name "*kernel_trace.csv" -exec cp {} $OUT/force_dist_r8_kernel_trace.csv \;
rm -rf $OUT/prof_dist
python -m pytest tests/test_gpu_dist.py -x -q -m gpu > $OUT/t_dist.log 2>&1

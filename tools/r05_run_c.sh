#!/bin/bash
# round 5, run C: parity of the segmented sort, a rank's compute as rank of 8 by tile / split variant, phase clocks of the in-LDS pass
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05c
mkdir -p $OUT
python -m pytest tests/test_gpu_segmented_sort.py -x -q -m gpu > $OUT/t_seg.log 2>&1
python -m pytest tests/test_gpu_lds_finish.py -x -q -m gpu > $OUT/t_fin.log 2>&1
B="python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1"
$B --as-rank-of 8 > $OUT/fd27_r8_geo5.json 2> $OUT/fd27_r8_geo5.err
GLU_HIP_SEG_MAX_GEO=4 $B --as-rank-of 8 > $OUT/fd27_r8_split2x2.json 2> $OUT/fd27_r8_split2x2.err
GLU_HIP_SEG_MAX_GEO=4 GLU_HIP_SEG_SPLIT_GEO=4 $B --as-rank-of 8 > $OUT/fd27_r8_split1x4.json 2> $OUT/fd27_r8_split1x4.err
GLU_HIP_SEG_LDS_FINISH=0 $B --as-rank-of 8 > $OUT/fd27_r8_off.json 2> $OUT/fd27_r8_off.err
$B > $OUT/fd27.json 2> $OUT/fd27.err
$B --as-rank-of 4 > $OUT/fd27_r4.json 2> $OUT/fd27_r4.err
$B --as-rank-of 2 > $OUT/fd27_r2.json 2> $OUT/fd27_r2.err
tools/finish_stamps_bench 28 4 > $OUT/finish_stamps_u32.txt 2>&1
tools/finish_stamps_bench 28 8 16 > $OUT/finish_stamps_u64_rank16.txt 2>&1
tools/finish_stamps_bench 28 8 24 > $OUT/finish_stamps_u64_rank24.txt 2>&1
tools/finish_stamps_bench 28 8 48 > $OUT/finish_stamps_u64_rank48.txt 2>&1
python bench.py --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python tools/sort_loop.py --log2 28 --key-bytes 8 > $OUT/c5_loop.txt 2>&1

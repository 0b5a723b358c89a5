#!/bin/bash
# round 5, run F: tie repair in two phases (parity), 64-bit in-LDS pass, segmented scatter 8 against 10 pairs per thread (same box,
# alternating), the spill-free 64-bit scatter
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05f
mkdir -p $OUT
python -m pytest tests/test_gpu_lds_finish.py -x -q -m gpu > $OUT/t_fin.log 2>&1
tools/finish_stamps_bench 28 8 16 > $OUT/finish_stamps_u64_rank16.txt 2>&1
tools/finish_stamps_bench 28 8 24 > $OUT/finish_stamps_u64_rank24.txt 2>&1
python tools/sort_loop.py --log2 28 --key-bytes 8 > $OUT/c5_loop.txt 2>&1
B="python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1 --no-verify"
for rep in 1 2 3; do
  GLU_HIP_SEG_LDS_FINISH=0 $B > $OUT/seg8_3pass_$rep.json 2> $OUT/err.txt
  GLU_HIP_LIB_PATH=$R/gl-radix-sort_amd/lib/libglu_hip_seg10.so GLU_HIP_SEG_LDS_FINISH=0 $B > $OUT/seg10_3pass_$rep.json 2> $OUT/err.txt
  $B > $OUT/seg8_lds_$rep.json 2> $OUT/err.txt
  GLU_HIP_LIB_PATH=$R/gl-radix-sort_amd/lib/libglu_hip_seg10.so $B > $OUT/seg10_lds_$rep.json 2> $OUT/err.txt
done
python -m pytest tests/test_gpu_segmented_sort.py tests/test_gpu_radix_sort.py -x -q -m gpu > $OUT/t_seg_sort.log 2>&1

#!/bin/bash
# round 5, run D: segmented parity after the defaults changed; phase clocks and tile variants of the in-LDS pass
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05d
mkdir -p $OUT
python -m pytest tests/test_gpu_segmented_sort.py -x -q -m gpu > $OUT/t_seg.log 2>&1
tools/finish_stamps_bench 28 4 > $OUT/finish_stamps_u32.txt 2>&1
tools/finish_stamps_bench 28 8 16 > $OUT/finish_stamps_u64_rank16.txt 2>&1
tools/finish_stamps_bench 28 8 48 > $OUT/finish_stamps_u64_rank48.txt 2>&1
for v in fsb_512x9_w6 fsb_384x12_w5 fsb_384x12 fsb_1024x5 fsb_768x6 fsb_640x8 fsb_256x18; do
  tools/bin/$v 28 8 16 > $OUT/variant_$v.txt 2>&1
done
for v in fsb32_512x9 fsb32_256x18_w5; do
  tools/bin/$v 28 4 > $OUT/variant_$v.txt 2>&1
done

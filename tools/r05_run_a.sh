#!/bin/bash
# round 5, run A: parity of the new in-LDS paths, then timings (C5 by rank bits; a rank's compute at world 1 and as rank of 8)
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05a
mkdir -p $OUT
python -m pytest tests/test_gpu_segmented_sort.py -x -q -m gpu > $OUT/t_seg.log 2>&1
python -m pytest tests/test_gpu_lds_finish.py -x -q -m gpu > $OUT/t_fin.log 2>&1
for rb in 16 24 48; do
  GLU_HIP_FINISH_RANK_BITS=$rb python tools/sort_loop.py --log2 28 --key-bytes 8 > $OUT/c5_rank_bits_$rb.txt 2>&1
done
python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1 > $OUT/fd27.json 2> $OUT/fd27.err
python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1 --as-rank-of 8 > $OUT/fd27_r8.json 2> $OUT/fd27_r8.err
GLU_HIP_SEG_SPLIT_GEO=4 python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1 --as-rank-of 8 > $OUT/fd27_r8_geo4.json 2> $OUT/fd27_r8_geo4.err
GLU_HIP_SEG_LDS_FINISH=0 python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1 --as-rank-of 8 > $OUT/fd27_r8_off.json 2> $OUT/fd27_r8_off.err
python bench.py --force-dist --log2-keys 27 --no-cpu-baseline --no-one-gpu --pipeline-depth 1 --as-rank-of 4 > $OUT/fd27_r4.json 2> $OUT/fd27_r4.err
python -m pytest tests/test_gpu_dist.py -x -q -m gpu > $OUT/t_dist.log 2>&1

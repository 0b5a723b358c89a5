#!/bin/bash
# round 5, run E: long runs of the whole-key sort (parity), the 64-bit in-LDS pass with tie detection from registers
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05e
mkdir -p $OUT
python -m pytest tests/test_gpu_lds_finish.py -x -q -m gpu > $OUT/t_fin.log 2>&1
tools/finish_stamps_bench 28 8 16 > $OUT/finish_stamps_u64_rank16.txt 2>&1
tools/bin/fsb_u64_waves1 28 8 16 > $OUT/finish_stamps_u64_rank16_waves1.txt 2>&1
python tools/sort_loop.py --log2 28 --key-bytes 8 > $OUT/c5_loop.txt 2>&1
python tools/sort_loop.py --log2 28 > $OUT/c3_loop.txt 2>&1
python tools/sort_loop.py --log2 28 --zeros 1 > $OUT/c3_zeros1_loop.txt 2>&1
python tools/sort_loop.py --log2 28 --zeros 0.01 > $OUT/c3_zeros001_loop.txt 2>&1
GLU_HIP_SORT_LONG_RUNS=0 python tools/sort_loop.py --log2 28 --zeros 1 > $OUT/c3_zeros1_loop_r4rule.txt 2>&1
python -m pytest tests/test_gpu_radix_sort.py -x -q -m gpu > $OUT/t_sort.log 2>&1

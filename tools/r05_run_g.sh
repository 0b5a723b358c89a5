#!/bin/bash
# round 5, run G: the whole GPU suite at this commit; the 4-bit / 64-bit scatter with 5 against 8 pairs per thread; bench
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05g
mkdir -p $OUT
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
for rep in 1 2 3; do
  python tools/sort_loop.py --log2 28 --key-bytes 8 --digit-bits 4 --steps 5 --warmup 2 > $OUT/u64_4bit_kpt8_$rep.txt 2>&1
  GLU_HIP_LIB_PATH=$R/gl-radix-sort_amd/lib/libglu_hip_u64_4bit_kpt5.so python tools/sort_loop.py --log2 28 --key-bytes 8 --digit-bits 4 --steps 5 --warmup 2 > $OUT/u64_4bit_kpt5_$rep.txt 2>&1
done
python bench.py --no-cpu-baseline > $OUT/bench_n1.json 2> $OUT/bench_n1.err

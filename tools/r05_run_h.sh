#!/bin/bash
# round 5, run H: the runs' key bits chosen on the device (parity), small-range keys on a fresh object, the suite
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05h
mkdir -p $OUT
python -m pytest tests/test_gpu_lds_finish.py -x -q -m gpu > $OUT/t_fin.log 2>&1
python tools/sort_loop.py --log2 28 --key-bits 28 --warmup 0 --steps 3 > $OUT/c3_28bit_fresh_object.txt 2>&1
python tools/sort_loop.py --log2 28 --key-bits 24 --warmup 0 --steps 3 > $OUT/c3_24bit_fresh_object.txt 2>&1
python tools/sort_loop.py --log2 28 > $OUT/c3_loop.txt 2>&1
python tools/sort_loop.py --log2 28 --key-bytes 8 > $OUT/c5_loop.txt 2>&1
python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_lds_finish.py > $OUT/pytest_gpu_rest.log 2>&1

#!/bin/bash
# Round 6: seven more minutes of each of the five fuzzers, new seeds, and the key distributions once more (run on the GPU box through gpurun).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06
mkdir -p $OUT
cd $R
timeout 460 python tools/fuzz.py 420 6106 > $OUT/fuzz_library_long.txt 2>&1
FUZZ_LARGE=1 timeout 460 python tools/fuzz.py 420 6107 > $OUT/fuzz_library_large_long.txt 2>&1
timeout 460 python tools/fuzz_one_object.py 420 6108 > $OUT/fuzz_one_object_long.txt 2>&1
timeout 460 python tools/fuzz_segments.py 420 6109 > $OUT/fuzz_segments_long.txt 2>&1
timeout 460 python tools/fuzz_heavy.py 420 6110 > $OUT/fuzz_heavy_long.txt 2>&1
python tools/measure_distributions_2p28.py > $OUT/distributions_2p28.txt 2>&1
tail -n 2 $OUT/fuzz_*_long.txt

#!/bin/bash
# Round 6: the GPU suite and the five fuzzers at HEAD (run on the GPU box through gpurun).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke_head.txt 2>&1
python -m pytest tests -q -m gpu > $OUT/pytest_gpu_head.txt 2>&1
timeout 260 python tools/fuzz.py 220 6006 > $OUT/fuzz_library.txt 2>&1
FUZZ_LARGE=1 timeout 200 python tools/fuzz.py 160 6007 > $OUT/fuzz_library_large.txt 2>&1
timeout 260 python tools/fuzz_one_object.py 220 6008 > $OUT/fuzz_one_object.txt 2>&1
timeout 200 python tools/fuzz_segments.py 160 6009 > $OUT/fuzz_segments.txt 2>&1
timeout 200 python tools/fuzz_heavy.py 160 6013 > $OUT/fuzz_heavy.txt 2>&1
tail -n 3 $OUT/smoke_head.txt $OUT/pytest_gpu_head.txt $OUT/fuzz_library.txt $OUT/fuzz_library_large.txt $OUT/fuzz_one_object.txt $OUT/fuzz_segments.txt $OUT/fuzz_heavy.txt

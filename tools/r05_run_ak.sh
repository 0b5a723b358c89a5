#!/bin/bash
# round 5, run AK: after the count kernels' peel mode with memory and the end of the back-off: smoke, the GPU suite, fuzzers, the distribution sweep, the reference-format ladder, the bench
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05ak
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
timeout 150 python tools/fuzz.py 120 9301 > $OUT/fuzz_library.txt 2>&1
FUZZ_LARGE=1 timeout 150 python tools/fuzz.py 120 9302 > $OUT/fuzz_library_large.txt 2>&1
timeout 150 python tools/fuzz_one_object.py 120 9303 > $OUT/fuzz_one_object.txt 2>&1
timeout 100 python tools/fuzz_segments.py 80 9304 > $OUT/fuzz_segments.txt 2>&1
python tools/measure_distributions_2p28.py > $OUT/distributions_2p28.txt 2>&1
tests/cpp/bin/bench_ladder > $OUT/bench_ladder_reference_format.txt 2>&1
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -n 2 $OUT/smoke.log; grep -n "passed\|failed\|FAILED" $OUT/pytest_gpu.log | tail -4; for f in $OUT/fuzz_*.txt; do tail -n 1 $f; done; cut -c1-120 $OUT/distributions_2p28.txt | tail -14; cut -c1-220 $OUT/bench_n1.json

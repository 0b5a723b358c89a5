#!/bin/bash
# round 5, run AC: the new thresholds of the attempt to end in LDS (u32 pairs 7 * 2^22, u64 3 * 2^21): ladders at the defaults, then smoke, the GPU suite, the bench
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05ac
mkdir -p $OUT
python tools/geometry_switch_ladder.py 24000000 40000000 1.04 pairs > $OUT/pairs_default.txt 2>&1
python tools/geometry_switch_ladder.py 5000000 9500000 1.06 u64 > $OUT/u64_default.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -n 2 $OUT/smoke.log; grep -n "passed\|failed\|FAILED" $OUT/pytest_gpu.log | tail -5; cut -c1-250 $OUT/bench_n1.json

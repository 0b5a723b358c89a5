#!/bin/bash
# round 5, run Y: keys-only ladder at the new defaults, then the whole GPU suite and the bench
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05y
mkdir -p $OUT
python tools/geometry_switch_ladder.py 6000000 70000000 1.07 keys > $OUT/keys_default.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
tail -n 3 $OUT/smoke.log; tail -n 3 $OUT/pytest_gpu.log; cut -c1-300 $OUT/bench_n1.json

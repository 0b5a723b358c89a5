#!/bin/bash
# round 5, run AA: (1) keys-only: attempt to end in LDS on / off, 30 .. 80 M keys; (2) pairs: the attempt from any size against the default, 24 .. 36 M;
# (3) a rank's compute of the sharded sort at 2^24 / 2^25 pairs per rank with non-temporal and with plain line stores
R=$GRAFT_REPO_ROOT
cd $R
OUT=$R/gpurun_out/r05aa
mkdir -p $OUT
python tools/geometry_switch_ladder.py 30000000 80000000 1.07 keys > $OUT/keys_finish_on.txt 2>&1
GLU_HIP_SORT_LDS_FINISH=0 python tools/geometry_switch_ladder.py 30000000 80000000 1.07 keys > $OUT/keys_finish_off.txt 2>&1
GLU_HIP_SORT_LDS_FINISH=0 GLU_HIP_SORT_LARGE_MIN=1000000000 python tools/geometry_switch_ladder.py 30000000 80000000 1.07 keys > $OUT/keys_finish_off_small.txt 2>&1
for L in 24 25; do
  for NT in 1 0; do
    GLU_HIP_SORT_NT_STORES=$NT python bench.py --force-dist --log2-keys $L --no-cpu-baseline --no-one-gpu --pipeline-depth 1 --as-rank-of 4 > $OUT/force_dist_2p${L}_nt${NT}.json 2> $OUT/fd.err
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r05aa/force_dist_*.json")):
    d=json.load(open(f)); print(f, d["ms_per_step"], d["phases_ms_rank0"])
PY

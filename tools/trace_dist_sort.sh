#!/bin/bash
# kernel timeline of the last sharded sort of `bench.py --force-dist` (one rank, real RCCL): start offset, duration, gap
# to the previous kernel's end (us).   bash tools/trace_dist_sort.sh [log2 pairs] [tag]      (GPU box)
L=${1:-27}; TAG=${2:-d}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_$TAG
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -o t -- python3 $R/bench.py --force-dist --log2-keys $L --steps 3 --warmup 2 --pipeline-depth 1 --no-verify --no-transport-fallback > /dev/null 2>&1
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/trace_$TAG/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# a sort starts at a count kernel (partition) that follows something that is not a glu kernel of the same sort: take the
# last ~40 kernels and cut at the last big gap
rows=rows[-60:]
cut=0
for i in range(1,len(rows)):
    if int(rows[i]["Start_Timestamp"])-int(rows[i-1]["End_Timestamp"])>200000: cut=i
rows=rows[cut:]
t0=int(rows[0]["Start_Timestamp"]); prev=None
for r in rows:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print("%8.1f %8.1f %7.1f  %s"%((s-t0)/1000,(e-s)/1000,((s-prev)/1000 if prev else 0),r["Kernel_Name"][:110]))
    prev=e
PY

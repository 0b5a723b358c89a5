#!/bin/bash
# kernel trace of the last of six sorts of N uniform pairs (start offset and duration of every launch, us)
#   bash tools/trace_one_sort.sh N [tag] [uniform|zero|two|runs64]     (GPU box; environment knobs of the library pass through)
N=${1:-6000000}; TAG=${2:-t}; KIND=${3:-uniform}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cat > /tmp/one_sort.py <<PY
import sys
sys.path.insert(0, "$R/gl-radix-sort_amd")
import numpy as np, glu_hip as G
m = $N
rng = np.random.default_rng(1)
kind = "$KIND"
if kind == "zero": keys = np.zeros(m, dtype=np.uint32)
elif kind == "two": keys = rng.integers(0, 2, m, dtype=np.uint32) * np.uint32(0x01010101)
elif kind == "runs64": keys = np.repeat(rng.integers(0, 2**32, (m + 63) // 64, dtype=np.uint32), 64)[:m].copy()
else: keys = rng.integers(0, 2**32, m, dtype=np.uint32)
vals = np.arange(m, dtype=np.uint32)
s = G.RadixSort(); s.prepare_internal_buffers(m)
for r in range(6):
    kb = G.ShaderStorageBuffer(keys); vb = G.ShaderStorageBuffer(vals); s(kb, vb, m, 0)
    kb.get_data(np.uint32)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_$TAG
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -o t -- python3 /tmp/one_sort.py > /dev/null 2>&1
cd $R
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/trace_$TAG/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=[i for i,r in enumerate(rows) if "count_kernel" in r["Kernel_Name"] or "single_block" in r["Kernel_Name"]]
# the last sort starts at the last count kernel that follows a non-glu kernel
starts=[i for i in last if i==0 or "glu_hip" not in rows[i-1]["Kernel_Name"]]
rows=rows[starts[-1]-1:]
t0=int(rows[0]["Start_Timestamp"])
for r in rows: print("%8.1f %8.1f  %s"%((int(r["Start_Timestamp"])-t0)/1000,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000,r["Kernel_Name"][:100]))
PY

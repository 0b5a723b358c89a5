# per-pass durations of the scatter kernel inside bench.py's sorts (rocprofv3 kernel trace); PPTAG names the output directory
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pp$PPTAG -o pp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-verify --no-alt > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob
import os
f=glob.glob("gpurun_out/pp"+os.environ.get("PPTAG","")+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "glu_hip" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# group into sorts: a sort starts at radix_pair_count ... collect scatter durations in order
sc=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000.0 for r in rows if "radix_scatter_lines_kernel" in r["Kernel_Name"]]
cnt=[(r["Kernel_Name"][:60],(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000.0) for r in rows if "count" in r["Kernel_Name"] or "unitsum" in r["Kernel_Name"]]
n=len(sc)//4
for p in range(4):
    v=[sc[4*i+p] for i in range(n)]
    print("scatter of pass %d: %s  avg %.1f us"%(p," ".join("%.0f"%x for x in v), sum(v)/len(v)))
print(cnt[:8])
PY

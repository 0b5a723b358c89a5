#!/bin/bash
# same-box A/B of two builds of libglu_hip.so: tools/ab_lib.sh <alt.so> [reps]  (bench.py sort time, scatter, count; C5 time)
ALT=$1; R=${2:-2}
for i in $(seq $R); do
  for lib in "" "$ALT"; do
    GLU_HIP_LIB_PATH=$lib python bench.py --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('${lib:-default}', 'sort', d['ms_per_step'], 'scatter', d['roofline']['avg_launch_ms'], 'count', d['roofline']['count_kernel_avg_ms'])"
    GLU_HIP_LIB_PATH=$lib python tools/u64_probe.py 2>/dev/null | head -2
  done
done

#!/bin/bash
# Chained (single-pass) scan: workgroup size x 16-byte groups per thread.  Builds tuning libraries HERE (no GPU needed):
#   bash tools/scan_chain_sweep.sh build
# and times them on the GPU box:   bash tools/scan_chain_sweep.sh run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CFGS="${CFGS:-1024x8 1024x4 512x8 512x4 512x16 256x8 256x16 256x4}"
if [ "$1" = build ]; then
  for c in $CFGS; do
    t=${c%x*}; g=${c#*x}
    (cd $R/gl-radix-sort_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wno-unused-function \
      -I$R/include -I. -DGLU_CHAIN_THREADS=$t -DGLU_CHAIN_GROUPS=$g -shared -o $R/gl-radix-sort_amd/lib/tuning_chain_$c.so glu_core.hip glu_hip.hip glu_sort_passes_u32.hip glu_sort_passes_u64.hip glu_sort_finish.hip glu_scan_reduce.hip) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
  done
  wait
  ls -la $R/gl-radix-sort_amd/lib/
else
  for c in $CFGS; do
    echo "== threads x groups = $c"
    GLU_HIP_LIB_PATH=$R/gl-radix-sort_amd/lib/tuning_chain_$c.so python $R/tools/scan_reduce_ladder.py 2>&1 | grep "^2^2[4-8]" | cut -c1-40
  done
fi

#!/bin/bash
# same-box A/B of the scratch placement tuning (sort_prepare tries six placements of the value scratch and keeps the fastest):
# alternating bench.py processes with GLU_HIP_SCRATCH_TUNE=1 (default) and =0.   bash tools/tune_ab.sh [rounds]
for r in $(seq ${1:-3}); do
  for t in 1 0; do
    GLU_VERBOSE=1 GLU_HIP_SCRATCH_TUNE=$t python bench.py --no-cpu-baseline --no-alt --steps 20 --warmup 5 2>/tmp/tune_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('tune=$t: sort %.4f ms  scatter %.4f ms  frac %.4f  whole %.4f  verified %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['whole_sort']['frac_of_peak_own_bytes'], d['verified']))"
    grep "scratch placement" /tmp/tune_err.txt | head -1
  done
done
for t in 1 0 1 0; do echo -n "C5 tune=$t: "; GLU_HIP_SCRATCH_TUNE=$t python tools/u64_probe.py 2>/dev/null | head -1; done

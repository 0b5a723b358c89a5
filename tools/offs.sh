#!/bin/bash
# per-array allocations with a chosen start offset (MiB) inside each: copy and scatter time over several processes
N=${1:-28}
shift
for o in "$@"; do
  echo -n "2^$N offs $o:"
  for i in 1 2 3 4 5 6 7 8; do SB_OFFS=$o SB_QUICK=1 ./tools/scatter_bench $N 2>&1 | grep -E "^carry" | sed "s/carry bits.*scatter \([0-9.]*\) ms.*/ \1/" | tr "\n" " "; done; echo
done

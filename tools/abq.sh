#!/bin/bash
# like ab64.sh for the 32-bit harness in SB_QUICK mode (production variant only)
R=$1; shift
declare -A RES
for i in $(seq $R); do
  for b in "$@"; do
    t=$(SB_QUICK=1 $b ${LOG2N:-28} ${ZERO:-0} 2>&1 | grep -E "^carry|^plain" | head -1 | sed 's/.*scatter \([0-9.]*\) ms.*/\1/')
    RES[$b]="${RES[$b]} $t"
  done
done
for b in "$@"; do echo "$b: $(echo ${RES[$b]} | tr ' ' '\n' | sort -n | tr '\n' ' ')"; done

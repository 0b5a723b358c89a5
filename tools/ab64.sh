#!/bin/bash
# alternate several harness binaries on one box, print min/median scatter ms of lines matching a pattern
# usage: tools/ab64.sh "<grep pattern>" reps bin1 bin2 ...
PAT=$1; R=$2; shift 2
declare -A RES
for i in $(seq $R); do
  for b in "$@"; do
    t=$($b 28 2>&1 | grep -E "$PAT" | head -1 | sed 's/.*scatter \([0-9.]*\) ms.*/\1/')
    RES[$b]="${RES[$b]} $t"
  done
done
for b in "$@"; do
  echo "$b: $(echo ${RES[$b]} | tr ' ' '\n' | sort -n | tr '\n' ' ')"
done

#!/bin/bash
# usage (GPU box): tools/ab_libs.sh <lib-a.so> <lib-b.so> [reps] [tune.py args]    — the same tune.py run on two builds of the library, alternating processes
A=$1; B=$2; N=${3:-2}; shift 3 2>/dev/null || shift 2
for i in $(seq 1 $N); do
  for L in $A $B; do
    python tools/tune.py --lib-path $L --variants 4 --rounds 3 --steps 5 "$@" 2>/dev/null | grep -E "^variant|^reads" | sed "s|^|$(basename $L) |"
  done
done

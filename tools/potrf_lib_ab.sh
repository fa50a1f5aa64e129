#!/bin/bash
# binary against binary on one box: potrf wall time (best / median of 9) with the library at $1 and the in-tree one, sizes $2...
base=$1; shift
for rep in 1 2; do
  for lib in "$base" ""; do
    echo "== GPMP_HIP_LIB=${lib:-<in-tree>}"
    GPMP_HIP_LIB=$lib timeout -k 10 120 python3 tools/potrf_ab.py GPMP_UNUSED_SWITCH 0 1 "$@" 2>&1 | grep "^n=" | grep "=     0:"
  done
done

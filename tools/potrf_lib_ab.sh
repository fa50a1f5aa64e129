#!/bin/bash
# binary against binary on one box: Cholesky wall time with the library at $1 (GPMP_HIP_LIB) and the in-tree one, sizes $2...
# (the way to A/B a change inside a kernel or a schedule constant: build the other variant into another .so)
base=$1; shift
for rep in 1 2; do
  for lib in "$base" ""; do
    for n in "$@"; do
      echo "== GPMP_HIP_LIB=${lib:-<in-tree>} n=$n"
      GPMP_HIP_LIB=$lib timeout -k 10 120 python3 tools/potrf_only.py $n 2>&1 | grep "^potrf ms"
    done
  done
done

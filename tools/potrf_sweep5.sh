#!/bin/bash
# potrf wall time: diagonal-first panel schedule on (default) / off
for n in 2048 4096 8192 16384 32768; do
  for df in 0 8192; do
    echo "n=$n GPMP_POTRF_DIAG_FIRST_BELOW=$df: $(GPMP_POTRF_DIAG_FIRST_BELOW=$df python3 tools/potrf_only.py $n 2>/dev/null | tail -1)"
  done
done

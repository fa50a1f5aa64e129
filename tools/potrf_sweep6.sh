#!/bin/bash
# potrf wall time: trailing update started behind the look-ahead update (default below 4096 rows left) or with it
for n in 2048 4096 8192 16384; do
  for v in 4096 0; do
    echo "n=$n GPMP_POTRF_MAIN_AFTER_LA_BELOW=$v: $(GPMP_POTRF_MAIN_AFTER_LA_BELOW=$v python3 tools/potrf_only.py $n 2>/dev/null | tail -1)"
  done
done

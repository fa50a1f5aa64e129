#!/bin/bash
for n in 2048 4096 8192 16384; do
  for cfg in "0 0 0" "0 0 100000" "100000 100000 100000" "100000 0 100000" "8192 4096 8192" "8192 0 8192" "4096 0 4096"; do
    set -- $cfg
    echo "n=$n W256_BELOW=$1 W128_BELOW=$2 MAIN_AFTER_LA_BELOW=$3: $(GPMP_POTRF_W256_BELOW=$1 GPMP_POTRF_W128_BELOW=$2 GPMP_POTRF_MAIN_AFTER_LA_BELOW=$3 python3 tools/potrf_only.py $n 2>/dev/null | tail -1)"
  done
done

#!/bin/bash
# potrf wall time with narrow panels in the tail (GPMP_POTRF_W256_BELOW / GPMP_POTRF_W128_BELOW = trailing size thresholds)
for n in 2048 4096 8192 16384; do
  for cfg in "0 0" "8192 0" "4096 0" "100000 0" "8192 4096" "8192 2048" "4096 2048" "100000 4096" "100000 100000" "6144 3072"; do
    set -- $cfg
    echo "n=$n W256_BELOW=$1 W128_BELOW=$2: $(GPMP_POTRF_W256_BELOW=$1 GPMP_POTRF_W128_BELOW=$2 python3 tools/potrf_only.py $n 2>/dev/null | tail -1)"
  done
done

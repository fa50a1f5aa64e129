#!/bin/bash
export GPMP_POTRF_W256_BELOW=4096 GPMP_POTRF_MAIN_AFTER_LA_BELOW=4096
for n in 2048 4096 8192 16384; do
  for cfg in "0 4096" "8 4096" "16 4096" "32 4096" "64 4096" "32 8192" "16 8192"; do
    set -- $cfg
    echo "n=$n TAIL_RESERVE_CUS=$1 TAIL_BELOW=$2: $(GPMP_POTRF_TAIL_RESERVE_CUS=$1 GPMP_POTRF_TAIL_BELOW=$2 python3 tools/potrf_only.py $n 2>/dev/null | tail -1)"
  done
done

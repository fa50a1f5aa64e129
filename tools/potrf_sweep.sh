#!/bin/bash
# potrf wall time at mid sizes under the panel-width / lean-kernel switches (tools/potrf_only.py prints two repetitions)
for n in 4096 8192 16384; do
  for wide in 4096 8192 16384 100000; do
    for lean in 4096 100000; do
      echo "n=$n GPMP_POTRF_WIDE_ABOVE=$wide GPMP_POTRF_LEAN_ABOVE=$lean: $(GPMP_POTRF_WIDE_ABOVE=$wide GPMP_POTRF_LEAN_ABOVE=$lean python3 tools/potrf_only.py $n 2>/dev/null | tail -1)"
    done
  done
done

#!/bin/bash
# A/B of the look-ahead Cholesky's switches (read at every call), each against its default in one process (tools/potrf_ab.py)
run() { timeout -k 10 150 python3 tools/potrf_ab.py "$@" 2>&1 | grep "^n="; }
run GPMP_POTRF_WIDE_ABOVE 4096 8192 8192 16384
run GPMP_POTRF_W256_BELOW 4096 8192 8192 16384
run GPMP_POTRF_LEAN_ABOVE 4096 8192 8192 16384
run GPMP_POTRF_LEAN_ABOVE 4096 2048 8192 16384
run GPMP_POTRF_MAIN_AFTER_LA_BELOW 4096 8192 8192 16384
run GPMP_POTRF_LA_SPLIT_ABOVE 8192 4096 8192 16384
run GPMP_POTRF_LA_SPLIT_ABOVE 8192 100000 16384 32768

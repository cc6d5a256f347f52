#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r04q
mkdir -p $OUT
IA3_DEBUG_TIMES=1 python scripts/host_stamps.py 2> $OUT/host_stamps.txt
tail -40 $OUT/host_stamps.txt

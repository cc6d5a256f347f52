#!/bin/bash
# generic GPU-box step: scripts/gpu_run.sh <tag> <command...> ; stdout+stderr -> gpurun_out/<tag>/log
set -e -o pipefail
OUT=gpurun_out/$1; shift
mkdir -p $OUT
"$@" > $OUT/log 2>&1 || { tail -40 $OUT/log; exit 1; }
tail -40 $OUT/log

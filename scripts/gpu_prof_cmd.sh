#!/bin/bash
# rocprofv3 kernel statistics of one python script: scripts/gpu_prof_cmd.sh <tag> <script.py> [args]  ->  gpurun_out/<tag>/kernel_stats.csv
set -e -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
S=$REPO/$1; shift
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o ks -- python3 $S "$@" > $OUT/stdout.log 2> $OUT/stderr.log || { tail -20 $OUT/stderr.log; exit 1; }
f=$(find $OUT/raw -name "ks_kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats.csv; rm -rf $OUT/raw
tail -5 $OUT/stdout.log | cut -c1-250
head -30 $OUT/kernel_stats.csv | cut -c1-150

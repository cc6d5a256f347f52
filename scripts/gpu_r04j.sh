#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r04j
mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT" -o ks -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --pool 1 > "$REPO/$OUT/bench_under_rocprof.json" 2> "$REPO/$OUT/ks.err"
cd "$REPO"
find $OUT -name "*kernel_trace.csv" | head -3
python scripts/step_gaps.py $(find $OUT -name "*kernel_trace.csv" | head -1) > $OUT/step_gaps.txt
tail -45 $OUT/step_gaps.txt

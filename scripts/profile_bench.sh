#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box (run from the repo root through gpurun):
#   scripts/profile_bench.sh <tag>  ->  gpurun_out/prof_<tag>/{ks_kernel_stats.csv, fetch_/write_counter_collection.csv, ...}
# kernel trace + stats in one pass; FETCH_SIZE and WRITE_SIZE in their own passes (never combined with tracing).
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$1
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ks -- python3 "$REPO/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-secondary --pool 1 > "$OUT/bench_under_rocprof.json" 2> "$OUT/ks.err"
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT" -o fetch -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --pool 1 > "$OUT/fetch.json" 2> "$OUT/fetch.err"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT" -o write -- python3 "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --pool 1 > "$OUT/write.json" 2> "$OUT/write.err"
echo "write done"
find "$OUT" -name "*.csv" | head -20

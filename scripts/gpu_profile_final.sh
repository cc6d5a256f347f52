#!/bin/bash
# final evidence of a round: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes over bench.py (scripts/profile_bench.sh),
# the SQ counters of the fit kernel (scripts/pmc_fit.sh if present), copied under gpurun_out/<tag> with the names
# profiles/make_traffic.py expects
set -e -o pipefail
TAG=${1:-r04z}
bash scripts/profile_bench.sh $TAG > gpurun_out/profile_$TAG.log 2>&1 || { tail -30 gpurun_out/profile_$TAG.log; exit 1; }
P=gpurun_out/prof_$TAG
mkdir -p gpurun_out/$TAG
f=$(find $P -name "ks_kernel_stats.csv" | head -1); cp "$f" gpurun_out/$TAG/kernel_stats_bench_steps5.csv
f=$(find $P -name "fetch_counter_collection.csv" | head -1); cp "$f" gpurun_out/$TAG/pmc_fetch_size.csv
f=$(find $P -name "write_counter_collection.csv" | head -1); cp "$f" gpurun_out/$TAG/pmc_write_size.csv
cp $P/bench_under_rocprof.json gpurun_out/$TAG/
rm -rf $P
ls -la gpurun_out/$TAG
head -12 gpurun_out/$TAG/kernel_stats_bench_steps5.csv | cut -c1-200

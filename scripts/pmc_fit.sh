#!/bin/bash
# SQ counter passes over one seed + fit pass (scripts/one_fov.py), counters only (no tracing):
#   scripts/pmc_fit.sh <tag>  ->  gpurun_out/pmc_<tag>/{a,b,c}_counter_collection.csv
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$1
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rocprofv3 -L > "$OUT/avail.txt" 2>&1 || true
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT" -o a -- python3 "$REPO/scripts/one_fov.py" > "$OUT/a.log" 2>&1
echo "pass a done"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT" -o b -- python3 "$REPO/scripts/one_fov.py" > "$OUT/b.log" 2>&1
echo "pass b done"
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d "$OUT" -o c -- python3 "$REPO/scripts/one_fov.py" > "$OUT/c.log" 2>&1 || echo "pass c failed (counter names?)"
echo "pass c done"
find "$OUT" -name "*.csv" | head

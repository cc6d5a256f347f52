#!/bin/bash
# rocprofv3 passes over the cubic warp (scripts/time_warp.py) on the GPU box, run from the repo root through gpurun:
#   scripts/profile_warp.sh <tag>  ->  gpurun_out/prof_<tag>/{warp_kernel_stats.csv, warp_fetch_/warp_write_counter_collection.csv}
# kernel trace + stats in one pass; FETCH_SIZE and WRITE_SIZE in their own passes (never combined with tracing).
set -e
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$1
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o warp -- python3 "$REPO/scripts/time_warp.py" > "$OUT/warp_time.log" 2> "$OUT/warp_ks.err"
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT" -o warp_fetch -- python3 "$REPO/scripts/time_warp.py" > /dev/null 2> "$OUT/warp_fetch.err"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT" -o warp_write -- python3 "$REPO/scripts/time_warp.py" > /dev/null 2> "$OUT/warp_write.err"
echo "write done"
if [ "$2" = "sq" ]; then
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT" -o warp_sqa -- python3 "$REPO/scripts/time_warp.py" > /dev/null 2> "$OUT/warp_sqa.err"
  echo "sq a done"
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT" -o warp_sqb -- python3 "$REPO/scripts/time_warp.py" > /dev/null 2> "$OUT/warp_sqb.err"
  echo "sq b done"
fi

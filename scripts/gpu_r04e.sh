#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r04e
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "voronoi or dependency_wait or daxprocesser or profiles_read or clustered_field or segmentation" > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
bash scripts/ab_fit2.sh lb3 "-DIA3_FIT_LB=3" 2>&1 | tee $OUT/ab_fit_lb3.log

#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r04p
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "movie_pipeline or batch" > $OUT/pytest.log 2>&1 || { tail -80 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log

#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r04i
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fork.py -m gpu -x -q -k "fit or refit or seidel or cluster or group or golden or legacy or centers" > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
python scripts/time_lone.py 2>&1 | tee $OUT/time_lone.log

#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r04d
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "seeds or batch or lazy or group" > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 600 python scripts/time_movies.py 12 $OUT/time_movies.json 2>&1 | tee $OUT/time_movies.log

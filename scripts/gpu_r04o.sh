#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r04o
mkdir -p $OUT
timeout -k 10 900 python scripts/time_movies.py 24 $OUT/time_movies.json 2>&1 | tee $OUT/time_movies.log
python scripts/time_lone.py 2>&1 | tee $OUT/time_lone.log

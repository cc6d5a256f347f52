#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r04f
mkdir -p $OUT
true
true
IA3_MOVIE_ONLY=1 timeout -k 10 600 python scripts/time_movies.py 12 $OUT/time_movies.json 2>&1 | tee $OUT/time_movies.log
python scripts/time_align.py 2>&1 | tee $OUT/time_align.log

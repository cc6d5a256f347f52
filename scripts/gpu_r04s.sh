#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r04s
mkdir -p $OUT
IA3_MOVIE_ONLY=1 IA3_WITH_TORCH=1 timeout -k 10 600 python scripts/time_movies.py 24 $OUT/time_movies_torch.json 2>&1 | tee $OUT/time_movies_torch.log
IA3_MOVIE_ONLY=1 timeout -k 10 600 python scripts/time_movies.py 24 $OUT/time_movies.json 2>&1 | tee $OUT/time_movies.log

#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r04c
mkdir -p $OUT
IA3_DEBUG_TIMING=1 IA3_MOVIE_ONLY=1 timeout -k 10 600 python scripts/time_movies.py 4 $OUT/time_movies.json > $OUT/time_movies.log 2> $OUT/time_movies.err || { tail -30 $OUT/time_movies.err; exit 1; }
grep -c "device finish" $OUT/time_movies.err || true
grep "dog_seed:" $OUT/time_movies.err | sort | uniq -c | sort -rn | head -20
tail -5 $OUT/time_movies.log

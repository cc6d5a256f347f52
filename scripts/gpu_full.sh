#!/bin/bash
# one GPU call: the whole GPU suite (with the carve-out counts printed), the two-rank rehearsal of bench.py on
# one device, the default bench line
set -e -o pipefail
OUT=gpurun_out/${1:-r04a}
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -s > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
grep -h "rows \|seeds .*overlapping\|equal-height" $OUT/pytest.log || true
IA3_BENCH_DEVICE=0 IA3_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-secondary --no-cpu-baseline > $OUT/bench_2rank_rehearsal.json 2> $OUT/bench_2rank_rehearsal.err || { tail -30 $OUT/bench_2rank_rehearsal.err; exit 1; }
cat $OUT/bench_2rank_rehearsal.json | cut -c 1-600
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -30 $OUT/bench.err; exit 1; }
cut -c 1-400 $OUT/bench.json

#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/r04g
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -k "align or drift or phase or dft" > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
python scripts/time_align.py 2>&1 | tee $OUT/time_align.log

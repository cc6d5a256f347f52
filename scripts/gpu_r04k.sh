#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/${1:-r04k}
mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fork.py -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -60 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err || { tail -20 $OUT/bench_quick.err; exit 1; }
python -c "
import json,sys
p=json.load(open('$OUT/bench_quick.json'))
print('ms_per_step', p['ms_per_step'], p['step_ms'], 'value', p['value'])
print(p['stage_ms_per_fov'])
print('fit frac', p['roofline']['frac'], p['roofline'].get('avg_launch_ms'))
"
REPO=$(pwd)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT" -o ks -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --pool 1 > "$REPO/$OUT/bench_under_rocprof.json" 2> "$REPO/$OUT/ks.err"
cd "$REPO"
python scripts/step_gaps.py $(find $OUT -name "*kernel_trace.csv" | head -1) > $OUT/step_gaps.txt
tail -19 $OUT/step_gaps.txt

#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/${1:-r04u}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "thread or batch or movie or pool or workspace or fit_fovs" > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
bash scripts/gpu_bench_only.sh $1
python - <<EOF
import json
p=json.load(open('$OUT/bench.json'))
c=p['secondary']['c5_movie']
print(c['scratch_cache'])
for i,r in enumerate(c['timeline_ms']): print(i, r)
EOF

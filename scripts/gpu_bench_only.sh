#!/bin/bash
set -e -o pipefail
OUT=gpurun_out/${1:-r04r}
mkdir -p $OUT
( time python bench.py ) > $OUT/bench.json 2> $OUT/bench.err || { tail -30 $OUT/bench.err; exit 1; }
tail -4 $OUT/bench.err
python - <<EOF
import json
p=json.load(open('$OUT/bench.json'))
s=p['secondary']
print('headline', p['ms_per_step'], p['step_ms']['median'], p['value'], 'frac', p['roofline']['frac'])
print('secondary keys', list(s.keys()))
if 'error' in s: print('ERROR', s['error'])
for k in ('forked_pool','u16_fov','clustered_float32','clustered_uint16'):
    if k in s: print(k, {a:b for a,b in s[k].items() if a not in ('workload','note')})
if 'c3_drift_fit' in s:
    c=s['c3_drift_fit']; print('c3', c['ms_per_fov'], c['align_ms'], c['fit_ms'], c['drifts'][:3], json.dumps(c['roofline'])[:600])
if 'c5_movie' in s:
    c=s['c5_movie']; print('c5', {a:b for a,b in c.items() if a not in ('workload','entry','stage_note','timeline_ms','warp')})
print('streaming', p['streaming']['f32']['frac_of_pcie_bound'], p['streaming']['u16']['frac_of_pcie_bound'])
print('cpu', p['cpu_baseline']['value'], p['cpu_baseline']['seconds'])
EOF

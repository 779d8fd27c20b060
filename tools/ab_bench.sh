#!/bin/bash
# same-box A/B of library builds: tools/ab_bench.sh libA.so libB.so ...   (prints ms/step and the stage split for each, twice)
for rep in 1 2; do
for lib in "$@"; do
  TVC_LIB_PATH=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stream 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_step']
print('$lib', round(d['ms_per_step'],3), 'filter', round(s['filter_net'],3), 'enc', round(s['encoder'],3), 'knn', round(s['knn'],3), {k[7:]:round(v,3) for k,v in s.items() if k.startswith('filter.')})"
done; done

#!/bin/bash
# bench.py alternately with tools/ab/old.so (HEAD) and the working-tree library, same box
for l in old new old new; do
  if [ $l = old ]; then export CCDM_LIB=$PWD/tools/ab/old.so; else unset CCDM_LIB; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$l', round(d['value'], 2), 'samples/s', round(d['ms_per_denoise_step'], 4), 'ms/denoise step')"
done

#!/bin/bash
# Same-box A/B of an environment switch: tools/ab_env.sh CCDM_NO_SUBPIXEL=1   (old = with the variable set, new = without)
for rep in 1 2; do
  for l in old new; do
    if [ $l = old ]; then v="env $1"; else v="env"; fi
    $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline ${AB_EXTRA} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', round(d['value'],2), 'samples/s', round(d['ms_per_step']/d['config']['denoise_steps_run'],4), 'ms/denoise step')"
  done
done

#!/bin/bash
# bench.py at several sub-batch stream counts (same box)
for s in ${@:-1 2 3 4}; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --substreams $s $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('substreams', $s, round(d['value'], 2), 'samples/s', round(d['ms_per_denoise_step'], 3), 'ms/denoise step', d['roofline']['avg_launch_ms'] if d.get('roofline') else None)"
done

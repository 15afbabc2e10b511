set -u
export TMPDIR=/tmp
for ss in 1 2 3 4 2; do
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --substreams $ss 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('substreams $ss', round(d['value'], 2), 'samples/s', round(d['ms_per_denoise_step'], 4), 'ms/denoise step')"
done

# single-stream throughput against the batch size: does the block-count quantisation of the big convs show?  (768 resident blocks: 64 tiles per sample at 128x128)
set -u
export TMPDIR=/tmp
for b in 24 32 36 48 60 64 72 96; do
  python bench.py --batch $b --substreams 1 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('batch $b', round(d['value'], 2), 'samples/s', round(d['ms_per_denoise_step'], 4), 'ms/denoise step', round(d['ms_per_denoise_step'] / $b * 1000, 2), 'us per sample-step')"
done

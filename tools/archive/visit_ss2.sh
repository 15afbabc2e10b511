# how many concurrent sub-batch streams pay at the Cityscapes-shaped configs (fewer, larger samples)?   SS="1 2 3 4" CFGS="c4 c5shard"
set -u
export TMPDIR=/tmp
for c in ${CFGS:-c4 c5shard c4b64}; do
for ss in ${SS:-1 2 1 2}; do
  python bench.py --config $c --substreams $ss --steps 1 --warmup 1 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$c substreams=$ss', round(d['value'], 4), 'samples/s', round(d['ms_per_denoise_step'], 4))
except Exception as ex:
    print('$c substreams=$ss failed', ex)"
done
done

# same-box A/B: tools/abx/old.so (built from HEAD or REV by tools/build_head_lib.sh) against the working-tree library
set -u
export TMPDIR=/tmp
EXTRA=${EXTRA:-}
for rep in 1 2; do
for l in old new; do
  if [ $l = old ]; then export CCDM_LIB=$PWD/tools/abx/old.so; else unset CCDM_LIB; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l', round(d['value'], 2), 'samples/s', round(d['ms_per_denoise_step'], 4), 'ms/denoise step')"
done
done

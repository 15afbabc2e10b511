# same-box comparison of several library builds:  LIBS="tools/ab/old.so tools/ab/p1.so -" bash tools/visit_libs.sh   ("-" = the working-tree library)
set -u
export TMPDIR=/tmp
EXTRA=${EXTRA:-}
for rep in 1 2; do
for l in $LIBS; do
  if [ "$l" = "-" ]; then unset CCDM_LIB; else export CCDM_LIB=$PWD/$l; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l', round(d['value'], 2), 'samples/s', round(d['ms_per_denoise_step'], 4), 'ms/denoise step')"
done
done

for b in 64 8; do for s in 0 16 32; do
  if [ $s = 0 ]; then unset CCDM_SLICES; else export CCDM_SLICES=$s; fi
  python bench.py --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch $b slices $s', round(d['value'],2), 'samples/s', round(d['ms_per_denoise_step'],4), 'ms/step')"
done; done

# visit r05k: the wave-per-phase Upsample kernel at every image size (tree) against its <= 1024-pixel rule (px1024): parity, then C4 / C5 shard / C2
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "upsample or c5 or c4" 2>&1 | tail -6
for cfg in c4 c2; do
for l in prev tree; do
  if [ $l = tree ]; then unset CCDM_LIB; else export CCDM_LIB=$PWD/tools/abx/$l.so; fi
  python bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --per-op gpurun_out/per_op_${cfg}_$l.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg', '$l', round(d['value'], 3), round(d['ms_per_denoise_step'], 3), 'single', round(d['single_stream']['value'], 3))"
done
python - <<PY
import json
a=json.load(open('gpurun_out/per_op_${cfg}_prev.json')); b=json.load(open('gpurun_out/per_op_${cfg}_tree.json'))
for x,y in zip(a,b):
    if abs(x['mean_us']-y['mean_us'])>0.06*x['mean_us']: print('  ', x['op'], x['name'], x['shape'], round(x['mean_us'],1), '->', round(y['mean_us'],1))
PY
done

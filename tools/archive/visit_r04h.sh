# the producer/consumer conv experiment per layer (CCDM_PC=1, experiments build): does it pay anywhere below 128x128?  + RCCL single-rank test
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "rccl or two_ranks" 2>&1 | tail -5
export CCDM_LIB=$PWD/tools/ab/exp.so
for l in old new; do
  if [ $l = new ]; then export CCDM_PC=1; else unset CCDM_PC; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-op gpurun_out/per_op_ab_$l.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$l', round(d['value'], 2), round(d['ms_per_denoise_step'],4), d['per_stage_us'], 'single', round(d['single_stream']['value'],2))"
done
python - <<'PY'
import json
a=json.load(open('gpurun_out/per_op_ab_old.json')); b=json.load(open('gpurun_out/per_op_ab_new.json'))
for x,y in zip(a,b):
    if abs(x['mean_us']-y['mean_us'])>0.04*x['mean_us']: print(x['op'], x['name'], x['shape'], round(x['mean_us'],1), '->', round(y['mean_us'],1))
PY

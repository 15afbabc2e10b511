# k_conv_ks with several n-tiles per block at 16x16: parity, then same-box A/B of the per-stage sums (CCDM_KS_MAX_AREA=128 = the round-3 routing)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "few_pixel or test_conv or fused_skip or unet or trajectory or shard" 2>&1 | tail -15 | tee gpurun_out/pytest_r04f.log
export CCDM_LIB=$PWD/tools/ab/exp.so
for l in old new old new; do
  if [ $l = old ]; then export CCDM_KS_MAX_AREA=128; else unset CCDM_KS_MAX_AREA; fi
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
unset CCDM_KS_MAX_AREA
for op in 20 52 53 15; do
  CCDM_TIMELINE_KS=1 CCDM_LIB=$PWD/tools/ab/abl.so CCDM_TIMELINE_OP=$op timeout 300 python tools/timeline_op.py 2>&1 | tail -1
done | tee gpurun_out/timeline_r04f.txt

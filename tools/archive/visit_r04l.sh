# head conv + epilogue in one launch (ccdm_head.hip): parity, full suite, then the bench with the per-op table
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "head_conv_fused or stem_conv" 2>&1 | tail -25
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for rep in 1 2; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-op gpurun_out/per_op_head.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 2), round(d['ms_per_denoise_step'],4), d['per_stage_us'], 'single', round(d['single_stream']['value'],2), round(d['single_stream']['ms_per_denoise_step'],4), 'dominant', round(d['roofline']['frac'],4), d['roofline']['launches_per_denoise_step'])"
done
python - <<'PY'
import json
a=json.load(open('gpurun_out/per_op_head.json'))
for x in a[:2]+a[-3:]: print(x['op'], x['name'], x['shape'], round(x['mean_us'],1), round(x['hbm_frac'],3))
PY

# stem kernel (one-hot on load, (tap, channel) K axis): parity, then the per-op table
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for rep in 1 2; do
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-op gpurun_out/per_op_stem.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'], 2), round(d['ms_per_denoise_step'],4), d['per_stage_us'], 'single', round(d['single_stream']['value'],2), 'dominant', round(d['roofline']['frac'],4), d['roofline']['launches_per_denoise_step'])"
done
python - <<'PY'
import json
a=json.load(open('gpurun_out/per_op_stem.json'))
for x in a[:3]+a[-2:]: print(x['op'], x['name'], x['shape'], round(x['mean_us'],1), round(x['hbm_frac'],3))
PY

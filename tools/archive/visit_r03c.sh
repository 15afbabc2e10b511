set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "few_pixel or test_conv or fused_skip" 2>&1 | tail -40 > gpurun_out/pytest_r03c.log
tail -30 gpurun_out/pytest_r03c.log
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-op gpurun_out/per_op_r03c.json > gpurun_out/bench_r03c.json 2> gpurun_out/bench_r03c.err
python -c "
import json; d=json.load(open('gpurun_out/bench_r03c.json')); print(d['value'], d['ms_per_denoise_step'], d['per_stage_us'], d['single_stream']['value'])" || tail -20 gpurun_out/bench_r03c.err
python - <<'PY'
import json
for o in json.load(open('gpurun_out/per_op_r03c.json')):
    if '8x8' in o['shape'] or o['kind']!='conv': print(o['op'], o['name'], o['shape'], round(o['mean_us'],1))
PY

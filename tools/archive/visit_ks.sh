set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${TAG:-x}
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "few_pixel" 2>&1 | tail -15
bash tools/visit_tl.sh
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-op gpurun_out/per_op_$TAG.json > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python -c "
import json; d=json.load(open('gpurun_out/bench_$TAG.json')); print(d['value'], d['ms_per_denoise_step'], d['per_stage_us'], d['single_stream']['value'])" || tail -20 gpurun_out/bench_$TAG.err

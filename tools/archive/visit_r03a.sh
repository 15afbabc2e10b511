set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for op in 33 39 20 13 24 51 7 16 29 52 65; do
  CCDM_LIB=$PWD/tools/ab/abl.so CCDM_TIMELINE_OP=$op timeout 300 python tools/timeline_op.py 2>&1 | tail -1
done > gpurun_out/timeline_r03a.txt
cat gpurun_out/timeline_r03a.txt
timeout 900 python bench.py --steps 3 --warmup 1 --per-op gpurun_out/per_op_r03a.json > gpurun_out/bench_eager_r03a.json 2> gpurun_out/bench_eager_r03a.err
python -c "
import json; d=json.load(open('gpurun_out/bench_eager_r03a.json')); print(d['value'], d['ms_per_denoise_step'], d['per_stage_us'], d['substreams2']['value'])"

export TMPDIR=/tmp CCDM_LIB=$PWD/tools/ab/exp.so
for d in 0 1 2 4 3 6 7; do
CCDM_RAW_DBG=$d python bench.py --steps 1 --warmup 1 --no-cpu-baseline --denoise-steps 20 --per-op gpurun_out/po.json >/dev/null 2>&1; python -c "
import json
a=json.load(open('gpurun_out/po.json')); print('dbg $d', [round(a[i]['mean_us'],1) for i in (5,10)])"
done

# anti-phased co-execution probe: stream A runs only the full-resolution (bulk) ops, stream B only the low-resolution ones
set -u
export TMPDIR=/tmp
export CCDM_LIB=$PWD/tools/ab/exp.so
run() {
  local tag=$1; shift
  CCDM_SKIP_OPS="$1" CCDM_SKIP_OPS_B="$2" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --substreams 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', round(d['ms_per_denoise_step'], 4), 'ms/step')"
}
BULK="0-9,71-85"; LOW="10-70"; ALL="0-85"
run "both-full           " "" ""
run "A=bulk  B=low       " "$LOW" "$BULK"
run "A=bulk  B=nothing   " "$LOW" "$ALL"
run "A=none  B=low       " "$ALL" "$BULK"
run "A=bulk  B=bulk      " "$LOW" "$LOW"
run "A=low   B=low       " "$BULK" "$BULK"
run "A=full  B=nothing   " "" "$ALL"
run "A=full  B=low       " "" "$BULK"
run "A=full  B=bulk      " "" "$LOW"

# does a half batch's bulk run at full efficiency with more (shorter-lived) workgroups per sample?  CCDM_SLICES (experiments build) overrides the
# 128x128 stage's slice count (12 by rule); stream A = bulk only / B = low only as in visit_r04d
set -u
export TMPDIR=/tmp
export CCDM_LIB=$PWD/tools/ab/exp.so
run() {
  local tag=$1; shift
  CCDM_SKIP_OPS="$1" CCDM_SKIP_OPS_B="$2" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --substreams $3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('slices=${CCDM_SLICES:-12}', '$tag', round(d['ms_per_denoise_step'], 4), 'ms/step')"
}
BULK="0-9,71-85"; LOW="10-70"; ALL="0-85"
for sl in 12 16 24 32; do
  export CCDM_SLICES=$sl
  run "A=bulk B=nothing (2 streams)" "$LOW" "$ALL" 2
  run "A=bulk B=low     (2 streams)" "$LOW" "$BULK" 2
  run "both full        (2 streams)" "" "" 2
  run "both full        (1 stream) " "" "" 1
  run "all full         (3 streams)" "" "" 3
done

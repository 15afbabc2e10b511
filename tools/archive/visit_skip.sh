# sensitivity of the step time to each stage: leave a stage's ops out (CCDM_SKIP_OPS, experiments build, garbage results) and time the
# product default (two sub-batch streams, graph replay) and the single-stream eager form
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
export CCDM_LIB=$PWD/tools/ab/exp.so
run() {
  local tag=$1; shift
  CCDM_SKIP_OPS="$1" python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', 'default', round(d['ms_per_denoise_step'], 4))"
  CCDM_SKIP_OPS="$1" python bench.py --steps 2 --warmup 1 --graph 0 --substreams 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', 'single-eager', round(d['ms_per_denoise_step'], 4))"
}
run none ""
run no8x8 "24-50"
run no16x16 "15-23,51-63"
run no32x32 "10-14,64-70"
run no64x64 "5-9,71-77"
run no128enc "1-4"
run no128dec "79-84"
run nolow "10-70"
run none2 ""

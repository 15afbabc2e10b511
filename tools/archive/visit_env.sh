# per-stage sums under environment switches of a CCDM_EXPERIMENTS build (tools/ab/exp.so):  ENVS="A=1;B=1 C=1" bash tools/visit_env.sh
set -u
export TMPDIR=/tmp
export CCDM_LIB=$PWD/tools/ab/exp.so
IFS=';' read -ra SETS <<< "${ENVS:-}"
for e in "" "${SETS[@]}" ""; do
  env $e python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$e]', round(d['value'], 2), d['per_stage_us'], 'single', round(d['single_stream']['value'],2))"
done

# same-box per-op A/B of several library builds: LIBS="tree bd4 bd6" OPS="64 71" bash tools/visit_ab_ops.sh
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
for l in ${LIBS:-tree}; do
  if [ $l = tree ]; then unset CCDM_LIB; else export CCDM_LIB=$PWD/tools/abx/$l.so; fi
  python bench.py --steps 1 --warmup 1 --no-cpu-baseline --per-op gpurun_out/per_op_ab_$l.json ${EXTRA:-} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
ops = json.load(open('gpurun_out/per_op_ab_$l.json'))
sel = [int(v) for v in '${OPS:-64 71}'.split()]
print('$l', round(d['value'], 2), 'single', round(d['single_stream']['value'], 2), d['per_stage_us'], {o: round(ops[o]['mean_us'], 1) for o in sel})"
done

#!/bin/bash
# Quick GPU visit: selected tests (-k expression) and a few bench lines.   gpurun -- 'bash tools/gpu_quick.sh TAG "<-k expr>" "<bench configs>"'
set -u
TAG=${1:-q}
KEXPR=${2:-}
CONFIGS=${3:-c2}
AB_ENV=${4:-}          # e.g. "CCDM_NO_PC=1": a second bench pass of every config with these variables set (same box A/B)
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "$KEXPR" -s 2>&1 | tail -80 > gpurun_out/pytest_quick_$TAG.log
  tail -60 gpurun_out/pytest_quick_$TAG.log
fi
for c in $CONFIGS; do
  extra="--steps 2 --warmup 1 --no-cpu-baseline"
  if [ -n "$AB_ENV" ]; then
    env $AB_ENV timeout 900 python bench.py --config $c $extra --no-secondary > gpurun_out/bench_${c}_${TAG}_B.json 2> gpurun_out/bench_${c}_${TAG}_B.err
    echo "== $c with $AB_ENV"; python -c "
import json; d=json.load(open('gpurun_out/bench_${c}_${TAG}_B.json')); print({k: d[k] for k in ('value','ms_per_denoise_step')}, 'roofline', (d.get('roofline') or {}).get('avg_launch_ms'))" || tail -5 gpurun_out/bench_${c}_${TAG}_B.err
  fi
  timeout 900 python bench.py --config $c $extra --per-op gpurun_out/per_op_${c}_$TAG.json > gpurun_out/bench_${c}_$TAG.json 2> gpurun_out/bench_${c}_$TAG.err
  echo "== $c"; python - "$c" "$TAG" <<'PY'
import json, sys
c, tag = sys.argv[1:3]
try:
    d = json.load(open(f"gpurun_out/bench_{c}_{tag}.json"))
    print({k: d[k] for k in ("value", "ms_per_denoise_step")}, "step frac", round(d["roofline_step"]["frac"], 4),
          "roofline", {k: round(v, 4) if isinstance(v, float) else v for k, v in (d.get("roofline") or {}).items() if k in ("frac", "frac_conv_io_only", "avg_launch_ms", "mfma_util", "share_of_denoise_step")})
    print("stages", d.get("per_stage_us"), "ss2", d.get("substreams2"))
    if d.get("roofline_attention"): print("attn", {k: v for k, v in d["roofline_attention"].items() if k in ("frac", "mfma_util", "avg_launch_ms")})
    for k, v in (d.get("roofline_shapes") or {}).items(): print("   ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
except Exception as e:
    print("ERR", e); print(open(f"gpurun_out/bench_{c}_{tag}.err").read()[-2000:])
PY
done

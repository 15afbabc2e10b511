# HIP runtime environment knobs against the headline command (none helps: GPU_MAX_HW_QUEUES, HIP_FORCE_DEV_KERNARG, DEBUG_CLR_GRAPH_PACKET_CAPTURE)
set -u
export TMPDIR=/tmp
for e in "${ENVS[@]:-}"; do :; done
IFS=';' read -ra SETS <<< "${ENVSETS:-;GPU_MAX_HW_QUEUES=8;GPU_MAX_HW_QUEUES=2;HIP_FORCE_DEV_KERNARG=0}"
for e in "${SETS[@]}"; do
  env $e python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary ${EXTRA:-} 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('[$e] ${EXTRA:-}', round(d['value'], 2), 'samples/s', round(d['ms_per_denoise_step'], 4))
except Exception as ex:
    print('[$e] failed', ex)"
done

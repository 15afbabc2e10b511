# two-stream overlap vs block lifetime of the bulk kernels: latency slicing (shorter-lived workgroups) under 1 / 2 / 3 sub-batch streams
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', '->', round(d['value'], 2), 'samples/s', round(d['ms_per_denoise_step'], 4), 'ms/step')"
}
for sl in throughput latency; do
  for ss in 1 2 3 4; do
    run --slicing $sl --substreams $ss
  done
done
run --slicing latency --substreams 2 --graph 0
run --slicing throughput --substreams 2 --graph 0

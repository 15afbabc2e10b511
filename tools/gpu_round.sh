#!/bin/bash
# One GPU visit: parity tests, smoke, bench (C2 product default + single-stream eager + exact-fp32, C4, C5 shard, batch 8), rocprofv3 kernel trace.
# Run via gpurun from the repo root:  gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r03a [quick]'
set -u
TAG=${1:-r03}
MODE=${2:-full}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -60 > gpurun_out/pytest_gpu_$TAG.log
cp gpurun_out/parity_report.json gpurun_out/parity_report_$TAG.json 2>/dev/null
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
# the headline: product defaults (HIP-graph replay, automatic sub-batching), + the untimed single-stream tapped pass (roofline, per-op table)
timeout 900 python bench.py --steps 3 --warmup 1 --per-op gpurun_out/per_op_$TAG.json > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err
if [ "$MODE" = "full" ]; then
timeout 900 python bench.py --steps 3 --warmup 1 --graph 0 --substreams 1 --no-cpu-baseline --no-secondary > gpurun_out/bench_eager1_$TAG.json 2> gpurun_out/bench_eager1_$TAG.err
timeout 900 python bench.py --steps 2 --warmup 1 --prec f32 --no-cpu-baseline --no-secondary > gpurun_out/bench_f32_$TAG.json 2> gpurun_out/bench_f32_$TAG.err
timeout 900 python bench.py --config c4 --steps 2 --warmup 1 --per-op gpurun_out/per_op_c4_$TAG.json > gpurun_out/bench_c4_$TAG.json 2> gpurun_out/bench_c4_$TAG.err
timeout 900 python bench.py --config c4b64 --steps 1 --warmup 1 --no-secondary > gpurun_out/bench_c4b64_$TAG.json 2> gpurun_out/bench_c4b64_$TAG.err
timeout 1200 python bench.py --config c5shard --steps 1 --warmup 1 --per-op gpurun_out/per_op_c5_$TAG.json > gpurun_out/bench_c5shard_$TAG.json 2> gpurun_out/bench_c5shard_$TAG.err
timeout 900 python bench.py --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/bench_b8_$TAG.json 2> gpurun_out/bench_b8_$TAG.err
timeout 900 python bench.py --batch 8 --slicing latency --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/bench_b8lat_$TAG.json 2> gpurun_out/bench_b8lat_$TAG.err
timeout 600 python tools/range_report.py --out gpurun_out/range_report_c2_$TAG.json > gpurun_out/range_report_$TAG.log 2>&1
fi
# kernel trace of the single-stream eager form of the bench command: the form whose per-launch durations describe the kernels (the bench's
# own roofline taps are taken on the same form)
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --graph 0 --substreams 1 --no-cpu-baseline --no-secondary > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head; ls -la gpurun_out/prof_$TAG | head
tail -8 gpurun_out/pytest_gpu_$TAG.log; cat gpurun_out/smoke_$TAG.log | tail -3
for f in default eager1 f32 c4 c4b64 c5shard b8 b8lat; do echo "== $f"; cat gpurun_out/bench_${f}_$TAG.json 2>/dev/null | cut -c1-1200; tail -2 gpurun_out/bench_${f}_$TAG.err 2>/dev/null; done
exit 0

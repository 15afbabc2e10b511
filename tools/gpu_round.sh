#!/bin/bash
# One GPU visit: parity tests, smoke, bench (eager + graph), rocprofv3 kernel trace.  Run via gpurun from the repo root.
set -u
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/pytest_gpu_$TAG.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_eager_$TAG.json 2> gpurun_out/bench_eager_$TAG.err
timeout 900 python bench.py --steps 3 --warmup 1 --graph 1 --no-cpu-baseline > gpurun_out/bench_graph_$TAG.json 2> gpurun_out/bench_graph_$TAG.err
timeout 900 python bench.py --steps 2 --warmup 1 --prec f32 --no-cpu-baseline > gpurun_out/bench_f32_$TAG.json 2> gpurun_out/bench_f32_$TAG.err
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$TAG -name "*kernel_stats*" | head; ls -la gpurun_out/prof_$TAG | head
tail -5 gpurun_out/pytest_gpu_$TAG.log; cat gpurun_out/smoke_$TAG.log | tail -3; cat gpurun_out/bench_eager_$TAG.json; cat gpurun_out/bench_graph_$TAG.json; cat gpurun_out/bench_f32_$TAG.json; tail -3 gpurun_out/bench_eager_$TAG.err

#!/bin/bash
set -u
TAG=${1:-x}
export TMPDIR=/tmp
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/rocprof_summary.py gpurun_out/prof_$TAG/trace_kernel_trace.csv gpurun_out/prof_$TAG/summary.md && tail -1 gpurun_out/prof_$TAG/summary.md

#!/bin/bash
# Copy the judged summaries of one gpu_round.sh / pmc_conv.sh visit (gpurun_out/*_$TAG*) into profiles/ (tracked).
set -eu
TAG=${1:?tag}
G=gpurun_out; P=profiles
cp $G/bench_eager_$TAG.json $P/r01_bench_f16x3_eager.json
cp $G/bench_graph_$TAG.json $P/r01_bench_f16x3_graph.json
cp $G/bench_f32_$TAG.json $P/r01_bench_f32_eager.json
cp $G/prof_$TAG/trace_kernel_stats.csv $P/r01_rocprofv3_kernel_stats.csv
python tools/rocprof_summary.py $G/prof_$TAG/trace_kernel_trace.csv $P/r01_rocprofv3_kernel_summary.md
cp $G/pytest_gpu_$TAG.log $P/r01_pytest_gpu.txt
cp $G/smoke_$TAG.log $P/r01_smoke.txt
if [ -d $G/pmc_$TAG ]; then
  for d in sq1 sq2 tcc1 tcc2 grbm; do
    f=$(find $G/pmc_$TAG/$d -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $P/r01_pmc/${d}_counter_collection.csv
  done
  python tools/pmc_to_json.py $G/pmc_$TAG $P/r01_pmc_dominant_kernel.json
fi
echo "profiles/ refreshed from $TAG"

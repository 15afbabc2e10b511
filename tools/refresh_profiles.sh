#!/bin/bash
# Copy the judged summaries of one gpu_round.sh / pmc_conv.sh visit (gpurun_out/*_$TAG*) into profiles/ (tracked), named per round.
set -eu
TAG=${1:?tag}
R=${2:-r03}
G=gpurun_out; P=profiles
cpif() { [ -s "$1" ] && cp "$1" "$2" || true; }
cpif $G/bench_default_$TAG.json $P/${R}_bench_default.json
cpif $G/bench_eager1_$TAG.json $P/${R}_bench_eager_single_stream.json
cpif $G/range_report_c2_$TAG.json $P/${R}_range_report_c2.json
cpif $G/bench_f32_$TAG.json $P/${R}_bench_f32_eager.json
cpif $G/bench_c4_$TAG.json $P/${R}_bench_c4.json
cpif $G/bench_c4b64_$TAG.json $P/${R}_bench_c4b64.json
cpif $G/bench_c5shard_$TAG.json $P/${R}_bench_c5shard.json
cpif $G/bench_b8_$TAG.json $P/${R}_bench_batch8.json
cpif $G/bench_b8lat_$TAG.json $P/${R}_bench_batch8_latency_slicing.json
cpif $G/per_op_$TAG.json $P/${R}_per_op_c2.json
cpif $G/per_op_c4_$TAG.json $P/${R}_per_op_c4.json
cpif $G/per_op_c5_$TAG.json $P/${R}_per_op_c5shard.json
cpif $G/parity_report_$TAG.json $P/${R}_parity_report.json
cpif $G/prof_$TAG/trace_kernel_stats.csv $P/${R}_rocprofv3_kernel_stats.csv
[ -s $G/prof_$TAG/trace_kernel_trace.csv ] && python tools/rocprof_summary.py $G/prof_$TAG/trace_kernel_trace.csv $P/${R}_rocprofv3_kernel_summary.md
cpif $G/pytest_gpu_$TAG.log $P/${R}_pytest_gpu.txt
cpif $G/smoke_$TAG.log $P/${R}_smoke.txt
if [ -d $G/pmc_$TAG ]; then
  mkdir -p $P/${R}_pmc
  for d in sq1 sq2 tcc1 tcc2 grbm; do
    f=$(find $G/pmc_$TAG/$d -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $P/${R}_pmc/${d}_counter_collection.csv
  done
  python tools/pmc_to_json.py $G/pmc_$TAG $P/${R}_pmc_dominant_kernel.json
fi
echo "profiles/ refreshed from $TAG as $R"

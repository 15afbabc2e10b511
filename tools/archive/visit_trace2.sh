# kernel trace of the product default (graph replay, two sub-batch streams) and of the single-stream graph run; overlap analysis on the box
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr2 -o t -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/tr2.log 2>&1
tail -2 /tmp/tr2.log | cut -c1-300
f=$(find /tmp/tr2 -name "*kernel_trace.csv" | head -1); ls -la $f
python $R/tools/trace_overlap.py $f $R/gpurun_out/trace_overlap_2streams.json | head -120
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -o t -- python $R/bench.py --steps 1 --warmup 1 --substreams 1 --no-cpu-baseline --no-secondary > /tmp/tr1.log 2>&1
f=$(find /tmp/tr1 -name "*kernel_trace.csv" | head -1)
python $R/tools/rocprof_summary.py $f $R/gpurun_out/trace_single_graph_summary.md > /dev/null 2>&1; head -30 $R/gpurun_out/trace_single_graph_summary.md

# round-4 first visit: box baseline, block timelines of the 16x16 / 32x32 / 64x64 convs inside the real step, dispatch gaps
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --per-op gpurun_out/per_op_r04a.json > gpurun_out/bench_r04a.json 2> gpurun_out/bench_r04a.err
cut -c1-600 gpurun_out/bench_r04a.json
for op in 20 53 52 13 65 7 2; do
  CCDM_LIB=$PWD/tools/ab/abl.so CCDM_TIMELINE_OP=$op timeout 300 python tools/timeline_op.py 2>&1 | tail -1
done | tee gpurun_out/timeline_r04a.txt
for g in 0 1; do
  rm -rf /tmp/gaps_$g
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps_$g -o trace -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --denoise-steps 30 --graph $g --substreams 1 --no-cpu-baseline --no-secondary > /tmp/gaps_$g.log 2>&1)
  echo "== graph=$g"; python tools/trace_gaps.py /tmp/gaps_$g 300
done | tee gpurun_out/gaps_r04a.txt

# visit r04m: full GPU suite + C4 / C5 benches, per-op tables and PMC passes after the staged many-class epilogue
set -u
export TMPDIR=/tmp
TAG=r04m
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -60 > gpurun_out/pytest_gpu_$TAG.log
cp gpurun_out/parity_report.json gpurun_out/parity_report_$TAG.json 2>/dev/null
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_$TAG.log 2>&1
timeout 900 python bench.py --config c4 --steps 2 --warmup 1 --per-op gpurun_out/per_op_c4_$TAG.json > gpurun_out/bench_c4_$TAG.json 2> gpurun_out/bench_c4_$TAG.err
timeout 900 python bench.py --config c4b64 --steps 1 --warmup 1 --no-secondary > gpurun_out/bench_c4b64_$TAG.json 2> gpurun_out/bench_c4b64_$TAG.err
timeout 1200 python bench.py --config c5shard --steps 1 --warmup 1 --per-op gpurun_out/per_op_c5_$TAG.json > gpurun_out/bench_c5shard_$TAG.json 2> gpurun_out/bench_c5shard_$TAG.err
bash tools/pmc_bench.sh $TAG c4 2 > gpurun_out/pmc_${TAG}_c4.log 2>&1
bash tools/pmc_bench.sh $TAG c5shard 2 > gpurun_out/pmc_${TAG}_c5.log 2>&1
tail -4 gpurun_out/pytest_gpu_$TAG.log; tail -2 gpurun_out/smoke_$TAG.log
for f in c4 c4b64 c5shard; do echo "== $f"; cut -c1-400 gpurun_out/bench_${f}_$TAG.json; done
grep -h posterior gpurun_out/pmc_${TAG}_c4.log gpurun_out/pmc_${TAG}_c5.log

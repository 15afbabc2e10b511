#!/bin/bash
# The round's evidence in one visit: full GPU suite, smoke, every bench config with per-op tables, rocprofv3 kernel trace, PMC passes.
#   gpurun --timeout 3000 -- 'bash tools/visit_round.sh r06a'      then      bash tools/refresh_profiles.sh r06a r06
set -u
TAG=${1:?tag}
export TMPDIR=/tmp
bash tools/gpu_round.sh $TAG full
bash tools/pmc_bench.sh $TAG c2 8 > gpurun_out/pmc_${TAG}_c2.log 2>&1
bash tools/pmc_bench.sh $TAG c5shard 2 > gpurun_out/pmc_${TAG}_c5.log 2>&1
bash tools/pmc_bench.sh $TAG c4 2 > gpurun_out/pmc_${TAG}_c4.log 2>&1
timeout 900 python bench.py --config c3shard --steps 1 --warmup 1 --no-pmc > gpurun_out/bench_c3shard_$TAG.json 2> gpurun_out/bench_c3shard_$TAG.err
tail -3 gpurun_out/pmc_${TAG}_c2.log

# visit r05p: the round's evidence — full GPU suite, smoke, every bench config with per-op tables, rocprofv3 kernel trace, PMC passes (C2, C4, C5 shard)
set -u
export TMPDIR=/tmp
bash tools/gpu_round.sh r05p full
bash tools/pmc_bench.sh r05p c2 8 > gpurun_out/pmc_r05p_c2.log 2>&1
bash tools/pmc_bench.sh r05p c5shard 2 > gpurun_out/pmc_r05p_c5.log 2>&1
bash tools/pmc_bench.sh r05p c4 2 > gpurun_out/pmc_r05p_c4.log 2>&1
tail -3 gpurun_out/pmc_r05p_c2.log

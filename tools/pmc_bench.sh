#!/bin/bash
# rocprofv3 --pmc passes over bench.py itself (short strided run of the named config): per (kernel, grid) counters of every launch of
# the step.  Separate passes, --kernel-trace only (gpurun refuses --pmc mixed with other trace domains).
#   gpurun -- 'bash tools/pmc_bench.sh r02x c2'   ->  gpurun_out/pmcb_<tag>_<cfg>/{sq1,sq2,tcc1,tcc2,grbm}/ + pmc_bench_<cfg>_<tag>.json
set -u
TAG=${1:-r03}
CFG=${2:-c2}
STEPS=${3:-8}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcb_${TAG}_$CFG
mkdir -p $OUT
cd /tmp
run() { # name counters...
  name=$1; shift
  timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --denoise-steps $STEPS --steps 1 --warmup 1 --graph 0 --substreams 1 --no-cpu-baseline --no-secondary > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU
run sq3 SQ_LDS_IDX_ACTIVE
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $GRAFT_REPO_ROOT
python tools/pmc_bench_summary.py $OUT gpurun_out/pmc_bench_${CFG}_$TAG.json $((2 * STEPS))

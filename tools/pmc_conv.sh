#!/bin/bash
# PMC counters for the dominant conv shape (separate passes; no trace domains mixed in, as gpurun requires)
set -u
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
run() { # name counters...
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o p -- python $GRAFT_REPO_ROOT/tools/bench_conv.py 1 1 > $OUT/$name.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU
run tcc1 FETCH_SIZE
run tcc2 WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $GRAFT_REPO_ROOT
for d in sq1 sq2 tcc1 tcc2 grbm; do f=$(find $OUT/$d -name "*counter_collection.csv" | head -1); echo "== $d $f"; python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
try:
    for r in csv.DictReader(open(sys.argv[1])):
        if "k_conv" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
except Exception as e:
    print("ERR", e)
for k, v in agg.items():
    print(k, {c: (sum(x) / len(x), len(x)) for c, x in v.items()})
PY
done

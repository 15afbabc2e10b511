export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_dh; mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/tcc1 -o p -- python $GRAFT_REPO_ROOT/tools/bench_conv.py 1 1 > $OUT/tcc1.log 2>&1
cd $GRAFT_REPO_ROOT; f=$(find $OUT/tcc1 -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "k_conv" in r["Kernel_Name"]]
print("FETCH_SIZE mean KiB", sum(v)/len(v), "-> MB read", 2*sum(v)/len(v)*1024/1e6)
PY
tail -2 $OUT/tcc1.log

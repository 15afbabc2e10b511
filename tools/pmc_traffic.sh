#!/bin/bash
# Round 6: where do the fetched bytes of the dominant conv class go?  FETCH_SIZE / WRITE_SIZE of the stand-alone 128x128 layers
# (tools/bench_conv.py shapes 0 = 32->32, 1 = 64->32 concat) under the ablation build, one rocprofv3 pass per (shape, bits, counter):
#   bits 0 full kernel | 64 no halo | 3 loads + stores only (no commit, no MFMA: the copy skeleton = the calibration of the
#   FETCH_SIZE x 2 rule on THIS access pattern) | 67 skeleton without halo (must read exactly the input) | 11 loads only | 75 loads only, no halo
#   gpurun -- 'CCDM_LIB=$PWD/tools/abx/abl.so bash tools/pmc_traffic.sh r06'
set -u
TAG=${1:-r06}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmct_$TAG
mkdir -p $OUT
cd /tmp
for shape in 0 1; do
  for bits in 0 64 3 67 11 75; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      d=$OUT/s${shape}_b${bits}_$ctr
      PMCRUN=$shape,$bits timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/tools/bench_conv.py 1 > $d.log 2>&1
    done
  done
done
cd $GRAFT_REPO_ROOT
python - "$OUT" <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
res = collections.OrderedDict()
for shape in (0, 1):
    for bits in (0, 64, 3, 67, 11, 75):
        row = {}
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            fs = glob.glob(f"{out}/s{shape}_b{bits}_{ctr}/**/*counter_collection.csv", recursive=True)
            vals = []
            for f in fs:
                for r in csv.DictReader(open(f)):
                    if "k_conv" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                        vals.append(float(r["Counter_Value"]))
            row[ctr + "_KiB_mean"] = sum(vals) / len(vals) if vals else None
            row[ctr + "_n"] = len(vals)
        ts = glob.glob(f"{out}/s{shape}_b{bits}_FETCH_SIZE/**/*kernel_trace.csv", recursive=True)
        durs = []
        for f in ts:
            for r in csv.DictReader(open(f)):
                if "k_conv" in r["Kernel_Name"]:
                    durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        row["us_mean"] = sum(durs) / len(durs) if durs else None
        if row["FETCH_SIZE_KiB_mean"] is not None:
            row["read_MB_x2"] = row["FETCH_SIZE_KiB_mean"] * 1024 * 2 / 1e6
        if row["WRITE_SIZE_KiB_mean"] is not None:
            row["write_MB"] = row["WRITE_SIZE_KiB_mean"] * 1024 / 1e6
        res[f"shape{shape}_bits{bits}"] = row
        print(f"shape{shape}_bits{bits}", row)
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
PY

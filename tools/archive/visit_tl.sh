set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for op in ${OPS:-33 39 40 24}; do
  CCDM_TIMELINE_KS=${KS:-1} CCDM_LIB=$PWD/tools/ab/abl.so CCDM_TIMELINE_OP=$op timeout 300 python tools/timeline_op.py 2>&1 | tail -1
done | tee gpurun_out/timeline_${TAG:-x}.txt

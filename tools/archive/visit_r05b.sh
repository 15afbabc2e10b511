# visit r05b: the GPU suite again (r05a stopped at its one failure), then block timelines of the sub-roofline tail ops (general kernel stamps)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -30 > gpurun_out/pytest_gpu_r05b.log
tail -5 gpurun_out/pytest_gpu_r05b.log
for op in ${OPS:-13 65 66 64 71 78 5 6 10 12 72}; do
  CCDM_LIB=$PWD/tools/abx/abl.so CCDM_TIMELINE_OP=$op timeout 300 python tools/timeline_op.py 2>&1 | tail -1
done | tee gpurun_out/timeline_r05b.txt

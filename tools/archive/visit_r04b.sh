set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "range or absmax or philox or graph or overflow or trained_like or attention" 2>&1 | tail -30 > gpurun_out/pytest_r04b.log
tail -15 gpurun_out/pytest_r04b.log
bash tools/visit_skip.sh 2>&1 | tee gpurun_out/skip_r04b.txt

# visit r05g: the wave-per-phase Upsample kernel: parity, then per-op / step A/B against HEAD~ (tools/abx/old.so)
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "upsample or unet_step or trajectory or updown" 2>&1 | tail -15
sed -i 's#tools/ab/old.so#tools/abx/old.so#' tools/visit_ab_stage.sh
bash tools/visit_ab_stage.sh
bash tools/visit_ab.sh

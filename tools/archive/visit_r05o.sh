# visit r05o: 8x8-tile variant of the wave-per-phase Upsample kernel (parity + per-op A/B against the previous commit), stride-2 weight staging A/B
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "upsample or unet_step or trajectory" 2>&1 | tail -6
LIBS="prev tree s2off prev tree s2off" OPS="5 10 51" bash tools/visit_ab_ops.sh

set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "test_conv or fused_skip or blocks_vs or subpixel or unet_step or latency" 2>&1 | tail -8
bash tools/visit_ab.sh

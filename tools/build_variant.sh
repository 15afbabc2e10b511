#!/bin/bash
# Build the working-tree sources into tools/abx/<name>.so with extra hipcc flags (same-box A/B / ablation flavours that travel to the GPU
# box: tools/ab/ is .gpurunignore'd, tools/abx/ is not; *.so stays out of git either way).
#   tools/build_variant.sh abl -DCCDM_ABLATION        tools/build_variant.sh new
set -e
cd "$(dirname "$0")/.."
NAME=${1:?name}; shift
mkdir -p tools/abx
cd ccdm_stochastic_segmentation_amd/csrc
ls *.hip | xargs -P 8 -I{} hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC "$@" -I../../include -c {} -o /tmp/variant_${NAME}_{}.o
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/variant_${NAME}_*.o -o ../../tools/abx/$NAME.so
rm -f /tmp/variant_${NAME}_*.o
ls -la ../../tools/abx/$NAME.so

#!/bin/bash
# Rebuild the library with the attention kernel's K/V staging arrays as HIP float4 structs (the variant that once returned wrong rows
# in the last (sample, head) on some boxes) and run the 1000-launch stress test against it, then against the shipped build.
# Run on the GPU box from the repo root:  bash tools/ubench/attn_float4_repro.sh
set -u
cd "$(dirname "$0")/../.."
SRC=ccdm_stochastic_segmentation_amd/csrc
mkdir -p tools/ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -shared -DCCDM_ATTN_STAGE_FLOAT4=1 -Iinclude $SRC/*.hip -o tools/ab/attn_float4.so 2>&1 | grep -i error
for lib in tools/ab/attn_float4.so ""; do
  echo "== ${lib:-shipped build}"
  CCDM_LIB=${lib:+$PWD/$lib} timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "attention_stress or attention_core or attention_head_width" 2>&1 | tail -4
done

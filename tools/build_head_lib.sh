#!/bin/bash
# Build the library from the sources at git HEAD (or REV=<commit>) into tools/abx/old.so with the product's flags
# (same-box A/B against the working tree: tools/visit_ab.sh / CCDM_LIB=tools/abx/old.so)
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/oldsrc && mkdir -p /tmp/oldsrc/csrc /tmp/oldsrc/include tools/abx
for f in $(git ls-tree -r --name-only ${REV:-HEAD} ccdm_stochastic_segmentation_amd/csrc); do git show ${REV:-HEAD}:$f > /tmp/oldsrc/csrc/$(basename $f); done
git show ${REV:-HEAD}:include/ccdm_hip.h > /tmp/oldsrc/include/ccdm_hip.h
cd /tmp/oldsrc/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -shared -I../include *.hip -o "$OLDPWD/tools/abx/old.so" 2>&1 | grep -v warning | grep -i error || true
ls -la "$OLDPWD/tools/abx/old.so"

#!/bin/bash
# Device ISA of ONE conv instantiation (register/ISA experiments): tools/isa.sh CK "TH,TW,WAVES,MI,NI[,KSP]" out.s
CK=${1:-16}; GEO=${2:-8,32,4,2,1}; OUT=${3:-/tmp/conv_exp.s}
cd "$(dirname "$0")/../ccdm_stochastic_segmentation_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DCCDM_EXPERIMENT -DCCDM_EXPERIMENT_CK=$CK "-DCCDM_EXPERIMENT_GEO=$GEO" -I../../include --offload-device-only -S -o $OUT ccdm_conv.hip 2>&1 | grep -v "hip-link"
grep "\.vgpr_count\|\.sgpr_count\|spill_count" $OUT
awk '/^_ZN4ccdm6k_conv/{f=1} /s_endpgm/{f=0} f' $OUT > ${OUT%.s}_k.s

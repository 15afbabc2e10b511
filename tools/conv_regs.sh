#!/bin/bash
# Register / spill report of every conv instantiation (device-only compile of ccdm_conv.hip)
cd "$(dirname "$0")/../ccdm_stochastic_segmentation_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include --offload-device-only -S -o /tmp/conv_all.s ccdm_conv.hip > /tmp/conv_all.log 2>&1
grep -c "warning\|error" /tmp/conv_all.log
python - <<'PY'
import re
txt=open('/tmp/conv_all.s').read()
print("<PREC,CK,KS,STRIDE,TH,TW,WAVES,MI,NI,KSP>  sgpr_spill vgpr vgpr_spill lds")
for blk in txt.split("  - .agpr_count:")[1:]:
    name=re.search(r'\.name:\s+(\S+)', blk).group(1)
    if "k_conv" not in name: continue
    t=re.search(r'ILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)', name).groups()
    g=lambda k: re.search(r'\.'+k+r':\s+(\d+)', blk).group(1)
    print(",".join(t), g("sgpr_spill_count"), g("vgpr_count"), g("vgpr_spill_count"))
PY

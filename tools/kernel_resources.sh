#!/bin/bash
# prints name / VGPRs / SGPRs / scratch / LDS / occupancy of every kernel of csrc/*.hip
cd "$(dirname "$0")/../mitsuba-renderer_amd/csrc"
for f in ${@:-sampler film trace shade measure}; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -x hip -c $f.hip -o /tmp/k_$f.o -Rpass-analysis=kernel-resource-usage 2>&1
done | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n,v,s,sc,l,o; n=$3} /^VGPRs:/{v="VGPR="$2} /^TotalSGPRs/{s="SGPR="$2} /ScratchSize/{sc="scratch="$3} /LDS Size/{l="LDS="$4} /Occupancy/{o="occ="$4} END{print n,v,s,sc,l,o}'

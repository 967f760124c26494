#!/bin/bash
# usage: tools/kres.sh file.hip  -> per-kernel resource usage table
f=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kres.o 2>&1 \
 | grep -E "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n, v, a, s, sc, o, l; n=$3; v=a=s=sc=o=l=""} /^VGPRs:/{v="vgpr="$2} /^AGPRs:/{a="agpr="$2} /VGPRs Spill/{s="vspill="$3} /SGPRs Spill/{s=s" sspill="$3} /ScratchSize/{sc="scratch="$3} /Occupancy/{o="occ="$3} /LDS Size/{l="lds="$4} END{print n, v, a, s, sc, o, l}' | while read n rest; do echo "$(echo $n | c++filt | cut -c1-90) $rest"; done

#!/bin/bash
# builds A/B variants of conv3x3_ws.hip: scratch/lib_<tag>.so ; usage: build_ws_variants.sh "tag:flags" ...
set -e
cd /root/repo/dig_amd/csrc
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast $flags -c conv3x3_ws.hip -o /tmp/ws_$tag.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E " VGPRs:|VGPRs Spill|error" | sed 's/remark: //; s/\[-Rpass.*//' | tr '\n' ' '; echo $tag;
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/ws_$tag.o conv3x3.o gemm.o elem.o linattn.o linattn_fused.o attn.o stem.o vit.o lfca.o api.o -o /root/repo/scratch/lib_$tag.so ) &
done
wait

#!/bin/bash
# builds A/B variants of the library: scratch/lib_<tag>.so ; usage: build_variants.sh "tag:flags" ...
set -e
cd /root/repo/dig_amd/csrc
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast $flags -c conv3x3.hip -o /tmp/conv_$tag.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Occupancy|LDS Size|error" | paste - - - - | sed 's/remark: [^ ]* //g' | grep conv3x3_kernelIDF16b | cut -c1-260 ;
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/conv_$tag.o conv3x3_ws.o conv3x3_db.o gemm.o pointwise.o elem.o linattn.o linattn_fused.o attn.o stem.o vit.o lfca.o api.o -o /root/repo/scratch/lib_$tag.so ) &
done
wait

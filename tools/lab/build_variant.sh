#!/bin/bash
# A/B variants of one kernel source: tools/lab/build_variant.sh <src.hip> "tag:flags" ...  ->  build/lib_<tag>.so
set -e
mkdir -p /root/repo/build
cd /root/repo/dig_amd/csrc
src=$1; shift
stem=${src%.hip}
others=$(ls *.o | grep -v "^$stem.o$" | tr '\n' ' ')
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=fast $flags -c $src -o /tmp/v_$tag.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|Spill|error|LDS Size" | sed 's/.*remark: //; s/\[-Rpass.*//' | tr '\n' ' '; echo "<= $tag";
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/v_$tag.o $others -o /root/repo/build/lib_$tag.so ) &
done
wait

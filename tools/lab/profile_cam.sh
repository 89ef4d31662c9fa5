#!/bin/bash
# rocprofv3 kernel stats of the CAM path (BASELINE.json configs[2]): B = 32 at 224x224 and B = 8 at 368x496 (733 tokens).
set -e
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $root
for cfg in "32 224 224 b32_224" "8 368 496 b8_368x496"; do
  set -- $cfg
  out=gpurun_out/r03_cam_$4
  rm -rf $out && mkdir -p $out
  B=$1 H=$2 W=$3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/lab/bench_cam.py > $out/bench.txt 2>&1
  cp $(find $out -name '*kernel_stats.csv' | head -1) gpurun_out/r03_cam_$4_kernel_stats.csv
  cat $out/bench.txt | grep -v amdgpu.ids
  python3 tools/lab/cam_profile_summary.py gpurun_out/r03_cam_$4_kernel_stats.csv $1 20 $2 $3 1 | tee gpurun_out/r03_cam_$4_summary.txt
  find $out -name '*.csv' -size +2M -delete 2>/dev/null || true
done

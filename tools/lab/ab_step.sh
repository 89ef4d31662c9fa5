#!/bin/bash
# per-launch table of one step under several libraries, inside ONE gpurun call: tools/lab/ab_step.sh libA.so libB.so ...
for lib in "$@"; do
  tag=$(basename $lib .so)
  DIG_HIP_LIB=$lib timeout -k 10 120 python tools/lab/step_launches.py > gpurun_out/st_${tag}.txt 2>&1 || exit 1
done

#!/bin/bash
# A/B of conv3x3 library variants inside ONE gpurun call (boxes differ by up to 7 %): tools/lab/ab_conv.sh libA.so libB.so ...
# Runs tools/lab/conv_layers.py with each library twice (A B ... A B ...) and writes gpurun_out/ab_<tag>_<round>.txt
for r in 1 2; do
  for lib in "$@"; do
    tag=$(basename $lib .so)
    DIG_HIP_LIB=$lib timeout -k 10 120 python tools/lab/conv_layers.py > gpurun_out/ab_${tag}_$r.txt 2>&1 || exit 1
  done
done

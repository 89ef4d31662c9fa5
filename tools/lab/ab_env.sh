#!/bin/bash
# A/B of environment settings inside ONE gpurun call: tools/lab/ab_env.sh tag1:VAR=val tag2:VAR=val ...  (twice each, interleaved)
for r in 1 2; do
  for spec in "$@"; do
    tag=${spec%%:*}; kv=${spec#*:}
    env $kv timeout -k 10 120 python tools/lab/conv_layers.py > gpurun_out/ab_${tag}_$r.txt 2>&1 || exit 1
  done
done

#!/bin/bash
set -e
mkdir -p gpurun_out
out=gpurun_out/gpu1.txt
: > $out
for t in rst_full rst_nodma rst_noepi rst_none; do
  DIG_HIP_LIB=$PWD/scratch/lib_$t.so timeout -k 10 120 python scratch/rs_trace.py 2>&1 | grep -v amdgpu.ids >> $out
done
cat $out

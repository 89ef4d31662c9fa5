#!/bin/bash
set -e
mkdir -p gpurun_out
out=gpurun_out/gpu1.txt
: > $out
for rep in 1 2; do
for t in default pf1 pf3; do
  if [ $t = default ]; then lib=$PWD/dig_amd/csrc/libdig_hip.so; else lib=$PWD/scratch/lib_$t.so; fi
  DIG_HIP_LIB=$lib timeout -k 10 120 python scratch/bench_ws2.py 2>&1 | grep -v amdgpu.ids >> $out
done
done
for t in t_full t_pf3 nostage ns_pf3; do
  DIG_HIP_LIB=$PWD/scratch/lib_$t.so timeout -k 10 120 python scratch/ws_trace.py 2>&1 | grep -E "SUMMARY|block cycles" >> $out
  DIG_HIP_LIB=$PWD/scratch/lib_$t.so timeout -k 10 120 python scratch/ws_trace.py pro 2>&1 | grep -E "SUMMARY|block cycles" | sed 's/^/   pro /' >> $out
done
cat $out

#!/bin/bash
mkdir -p gpurun_out
out=gpurun_out/gpu1.txt
: > $out
timeout -k 10 240 python -u -m pytest tests/test_gpu_kernels.py -x -q --timeout 60 --timeout-method thread -p no:cacheprovider -k "register_stationary" 2>&1 | tail -5 >> $out || { cat $out; exit 1; }
timeout -k 10 120 python scratch/bench_rs.py 2>&1 | grep -v amdgpu.ids >> $out
for t in rst_d1 rst_d1b rst_d2 rst_d2late; do
DIG_HIP_LIB=$PWD/scratch/lib_$t.so timeout -k 10 120 python scratch/rs_trace.py 2>&1 | grep -v amdgpu.ids | tail -5 >> $out
done
cat $out

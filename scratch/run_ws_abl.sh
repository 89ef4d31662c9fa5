#!/bin/bash
set -e
mkdir -p gpurun_out
out=gpurun_out/ws_abl.txt
: > $out
echo "--- trace plain blk36" >> $out
DIG_HIP_LIB=$PWD/scratch/lib_trace36.so timeout -k 10 120 python scratch/ws_trace.py >> $out 2>&1
echo "--- trace pro blk36" >> $out
DIG_HIP_LIB=$PWD/scratch/lib_trace36.so timeout -k 10 120 python scratch/ws_trace.py pro >> $out 2>&1
cat $out

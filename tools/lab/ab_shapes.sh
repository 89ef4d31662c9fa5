#!/bin/bash
# per-layer time of every conv3x3_db launch at batch $1 under each forced block shape (and the cost model's own choice)
B=${1:-20}
for r in 1 2; do
  for th in auto 32 16 16x64 32x64; do
    if [ $th = auto ]; then timeout -k 10 120 python tools/lab/conv_layers.py $B > gpurun_out/ab_sh${th}_$r.txt 2>&1 || exit 1
    else DIG_DB_TILE=$th timeout -k 10 120 python tools/lab/conv_layers.py $B > gpurun_out/ab_sh${th}_$r.txt 2>&1 || exit 1; fi
  done
done

#!/bin/bash
# One-GPU batch sweep behind the predicted strong-scaling efficiency: per-GPU batch 32/N images for
# N = 1, 2, 4, 8 ranks, single timestep and the guidance loop's five batched timesteps (BASELINE.json configs[3]).
# usage (on the GPU box): bash tools/lab/batch_sweep.sh > gpurun_out/r02_batch_sweep.jsonl
set -e
for b in 32 16 8 4; do
  timeout -k 10 200 python3 bench.py --batch $b --steps 30 --warmup 5 --no-cpu-baseline
done
for b in 32 16 8 4; do
  timeout -k 10 300 python3 bench.py --batch $b --timesteps 5 --steps 10 --warmup 3 --no-cpu-baseline
done

#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --steps 30 --warmup 10 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err || { tail -20 gpurun_out/bench_a.err; exit 1; }
cat gpurun_out/bench_a.json
for b in 4 8 16; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --batch $b --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch', d['config']['per_gpu_batch'], d['value'], 'maps/s', d['ms_per_step'], 'ms')"
done
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --batch 4 --timesteps 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch 4x5', d['value'], 'maps/s', d['ms_per_step'], 'ms')"

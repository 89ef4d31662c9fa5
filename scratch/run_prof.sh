#!/bin/bash
set -e
mkdir -p gpurun_out/prof_r2a
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2a -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r2a/bench.json 2> gpurun_out/prof_r2a/bench.err
ls -R gpurun_out/prof_r2a | head -20
tail -2 gpurun_out/prof_r2a/bench.json

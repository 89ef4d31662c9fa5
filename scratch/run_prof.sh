#!/bin/bash
set -e
mkdir -p gpurun_out/prof_r2b
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2b -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/prof_r2b/bench.json 2> gpurun_out/prof_r2b/bench.err
ls -R gpurun_out/prof_r2b | head -20
tail -2 gpurun_out/prof_r2b/bench.json

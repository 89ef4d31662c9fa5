#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel trace + stats of bench.py, then the two PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes) for HBM traffic per launch.  usage: bash tools/profile_round.sh r02   (writes gpurun_out/<tag>_*)
set -e
tag=${1:-r02}
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/${tag}_trace $root/gpurun_out/${tag}_fetch $root/gpurun_out/${tag}_write
cd /tmp && export TMPDIR=/tmp
cd $root
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_trace/bench.json 2> gpurun_out/${tag}_trace/bench.err
echo trace done
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_fetch/bench.json 2> gpurun_out/${tag}_fetch/bench.err
echo fetch done
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/${tag}_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_write/bench.json 2> gpurun_out/${tag}_write/bench.err
echo write done
f=$(find gpurun_out/${tag}_fetch -name '*counter_collection.csv' | head -1)
w=$(find gpurun_out/${tag}_write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_traffic.py $f $w gpurun_out/${tag}_pmc_traffic.json
cp $(find gpurun_out/${tag}_trace -name '*kernel_stats.csv' | head -1) gpurun_out/${tag}_kernel_stats.csv
# the raw per-dispatch CSVs are tens of MB: keep only the summaries
find gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_trace -name '*.csv' -size +2M -delete 2>/dev/null || true
cat gpurun_out/${tag}_pmc_traffic.json | head -50
tail -c 600 gpurun_out/${tag}_trace/bench.json

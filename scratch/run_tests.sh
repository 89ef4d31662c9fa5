#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/tests.txt
rc=${PIPESTATUS[0]}
cat gpurun_out/tests.txt
exit $rc

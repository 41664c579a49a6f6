#!/bin/bash
# round 2, GPU call 1: full GPU suite (new full-size tests included) + the new bench line
set -x
mkdir -p gpurun_out/r2c1
cd $GRAFT_REPO_ROOT
nproc > gpurun_out/r2c1/nproc.txt; free -g >> gpurun_out/r2c1/nproc.txt
timeout 1500 python -m pytest tests -m gpu -q -rP --durations=15 > gpurun_out/r2c1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r2c1/pytest.log
timeout 600 python bench.py > gpurun_out/r2c1/bench.json 2> gpurun_out/r2c1/bench.err
echo "bench rc=$?" >> gpurun_out/r2c1/bench.err
tail -5 gpurun_out/r2c1/pytest.log
cat gpurun_out/r2c1/bench.json

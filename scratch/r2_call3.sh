#!/bin/bash
set -x
O=gpurun_out/r2c3; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_score.py tests/test_gpu_tile32.py tests/test_gpu_sampler.py tests/test_gpu_pipeline.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench.json 2>$O/bench.err
tail -3 $O/pytest.log; cat $O/bench.json

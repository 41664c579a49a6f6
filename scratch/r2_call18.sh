#!/bin/bash
O=gpurun_out/r2c18; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_coupled.py tests/test_gpu_sampler.py tests/test_gpu_pipeline.py tests/test_gpu_tile32.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | cut -c1-200

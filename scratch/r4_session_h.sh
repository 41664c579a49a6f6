#!/bin/bash
# round 4, session H: 64-row tiles of the score trunk
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4h; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_tile32.py tests/test_gpu_chain.py tests/test_gpu_score.py tests/test_gpu_pipeline.py > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 400 python scratch/chain_check.py > $O/plans.txt 2>&1; cat $O/plans.txt

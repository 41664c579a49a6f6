#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_edge_cases.py > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for B in 5 64 320; do timeout 100 python scratch/enc_profile.py $B 30 graph 2>/dev/null | tail -1; done
bash scratch/enc_kernel_stats.sh 320 $O/enc.txt > /dev/null 2>&1; grep "fps\|sum" $O/enc.txt

#!/bin/bash
# quick GPU check after an encoder-side change: operator / encoder parity tests, then the encoder's wall time and kernel statistics
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/quick; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_edge_cases.py tests/test_gpu_encoder.py tests/test_gpu_sa_paths.py -q -m gpu -p no:cacheprovider -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for mode in forward graph; do for B in 5 64 320 640; do timeout 100 python scratch/enc_profile.py $B 30 $mode 2>/dev/null | tail -1; done; done | tee $O/encoder_wall.txt
bash scratch/enc_kernel_stats.sh 320 $O/encoder320_kernel_stats.txt > /dev/null 2>&1; cat $O/encoder320_kernel_stats.txt

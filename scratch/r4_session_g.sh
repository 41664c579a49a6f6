#!/bin/bash
# round 4, session G: the whole GPU suite on the current tree, GroupAll remainder on the side stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; rm -rf $O; mkdir -p $O
for mode in forward graph; do for B in 64 320 448; do timeout 100 python scratch/enc_profile.py $B 30 $mode 2>/dev/null | tail -1; done; done > $O/encoder_wall.txt; cat $O/encoder_wall.txt
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=12 > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log

#!/bin/bash
# builds the timing lib beside the shipped one:  scratch/tune.sh   (then run scratch/tune_gpu.sh on the GPU box)
set -e
export GP_EXTRA_FLAGS="$TUNE_FLAGS"
cd /root/repo
GP_TIMING=1 python -m genpose_amd.build --force 2>&1 | grep -i "error" && exit 1
cp genpose_amd/lib/libgenpose_hip.so genpose_amd/lib/libgenpose_hip_timing.so
python -m genpose_amd.build --force 2>&1 | grep -i "error" && exit 1
echo built both

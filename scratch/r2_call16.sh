#!/bin/bash
O=gpurun_out/r2c16; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 200 python scratch/fps_time.py 5 64 320 2>&1 | grep -v amdgpu.ids > $O/out.txt; cat $O/out.txt
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_preprocess.py -m gpu -q 2>&1 | tail -1
timeout 200 python scratch/enc_profile.py 320 2>&1 | grep encoder; timeout 200 python scratch/enc_profile.py 64 2>&1 | grep encoder

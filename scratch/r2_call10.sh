#!/bin/bash
O=gpurun_out/r2c10; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in "" _noslp; do
  GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip$v.so timeout 200 python scratch/pc_time.py 64 128 320 2>&1 | grep -v amdgpu.ids >> $O/out.txt
  GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip$v.so timeout 200 python scratch/enc_profile.py 320 2>&1 | grep -v amdgpu.ids >> $O/out.txt
  GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip$v.so timeout 200 python scratch/enc_profile.py 64 2>&1 | grep -v amdgpu.ids >> $O/out.txt
done
timeout 600 python -m pytest tests/test_training.py tests/test_gpu_sampler.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
cat $O/out.txt

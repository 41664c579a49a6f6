#!/bin/bash
O=gpurun_out/r2c14; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in "" _nw4; do
  GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip$v.so timeout 200 python scratch/pc_time.py 128 192 256 320 640 2>&1 | grep -v amdgpu.ids >> $O/out.txt
done
cat $O/out.txt

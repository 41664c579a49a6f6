#!/bin/bash
O=gpurun_out/r2_ablation; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in "" _abl_EMPTY _abl_NOPRO _abl_NOEPI _abl_NOTAIL _abl_NOWLOAD _abl_CORE _abl_CORENW; do
  GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip$v.so timeout 200 python scratch/pc_time.py 64 128 320 2>&1 | grep -v amdgpu.ids >> $O/ablation.txt
done
cat $O/ablation.txt

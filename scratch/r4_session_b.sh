#!/bin/bash
# round 4, session B: encoder changes (ring-kernel biases in LDS, ball query centres per wave, deferred-join grouping) against the base library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_encoder.py tests/test_gpu_ops.py tests/test_gpu_sa_paths.py tests/test_gpu_pipeline.py \
   tests/test_gpu_fullsize.py::test_encoder_vs_oracle_at_bench_sizes tests/test_gpu_fullsize.py::test_drop_in_eval_single_as_timed > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_dist8.py > $O/pytest_dist8.log 2>&1; tail -3 $O/pytest_dist8.log
{
for B in 5 64 320; do
  GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip_base.so timeout 100 python scratch/enc_profile.py $B 30 forward 2>/dev/null | tail -1 | sed 's/^/base /'
  for mode in forward pass graph; do timeout 100 python scratch/enc_profile.py $B 30 $mode 2>/dev/null | tail -1; done
done
} > $O/enc_wall.txt; cat $O/enc_wall.txt
bash scratch/enc_kernel_stats.sh 320 $O/encoder320_kernel_stats.txt > /dev/null 2>&1; cat $O/encoder320_kernel_stats.txt

#!/bin/bash
# round 4, session A: the new tests (no -x: every failure is wanted), the drop-in leg, the encoder baseline
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_edge_cases.py tests/test_gpu_sa_paths.py tests/test_gpu_encoder.py tests/test_gpu_sampler.py \
  tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py::test_drop_in_eval_single_as_timed tests/test_gpu_fullsize.py::test_config2_full_pipeline_256 > $O/pytest_new.log 2>&1
tail -15 $O/pytest_new.log
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_dist8.py > $O/pytest_dist8.log 2>&1; tail -5 $O/pytest_dist8.log
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_bench.py --durations=10 > $O/pytest_bench.log 2>&1; tail -15 $O/pytest_bench.log
timeout 300 python bench.py --only-drop-in > $O/drop_in.json 2> $O/drop_in.err; cat $O/drop_in.json
bash scratch/enc_kernel_stats.sh 320 $O/encoder320_kernel_stats.txt > /dev/null 2>&1; tail -1 $O/encoder320_kernel_stats.txt
timeout 100 python scratch/enc_profile.py 64 20 2>/dev/null | tail -1
timeout 100 python scratch/enc_profile.py 5 50 2>/dev/null | tail -1

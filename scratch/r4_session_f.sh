#!/bin/bash
# round 4, session F: tracking frame with fewer small launches; likelihood chain test; bench line with the new legs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_pipeline.py tests/test_gpu_sampler.py tests/test_gpu_chain_vjp.py::test_likelihood_chain_vs_tile tests/test_gpu_edge_cases.py > $O/pytest.log 2>&1; tail -4 $O/pytest.log
timeout 100 python scratch/track_one.py 2>/dev/null | tail -1
timeout 300 python scratch/bench_tracking.py 16 64 128 2>/dev/null > $O/tracking.txt; cat $O/tracking.txt
timeout 200 python bench.py --no-cpu-baseline --no-secondary --tracking --sequences 1 --steps 30 --warmup 8 2>/dev/null | cut -c1-400

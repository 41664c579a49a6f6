#!/bin/bash
# round 4, session D: chain form of the forward + vector-Jacobian right-hand sides
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_chain_vjp.py tests/test_gpu_energy_sampling.py tests/test_gpu_chain.py tests/test_gpu_score.py > $O/pytest_vjp.log 2>&1; tail -25 $O/pytest_vjp.log
{
for rows in 3200 12800 32000 64000; do
  timeout 120 python scratch/vjp_time.py $rows 16 2>/dev/null | grep energy
  timeout 120 python scratch/vjp_time.py $rows 128 2>/dev/null | grep energy
done
timeout 120 python scratch/vjp_time.py 32000 0 2>/dev/null
} > $O/vjp_plans.txt; cat $O/vjp_plans.txt
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_sampler.py tests/test_gpu_tile32.py > $O/pytest_samplers.log 2>&1; tail -5 $O/pytest_samplers.log
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_quick.json 2>/dev/null; cut -c1-300 $O/bench_quick.json

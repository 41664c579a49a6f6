#!/bin/bash
# round 6, second GPU session: the new tests, plan timing, the whole GPU suite, the bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6s2; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_shared_plan.py tests/test_gpu_sampler.py -q -x -s -p no:cacheprovider > $O/new_tests_a.log 2>&1; tail -15 $O/new_tests_a.log
timeout 300 python scratch/ode_plan_time.py > $O/ode_plans.txt 2>&1; cat $O/ode_plans.txt
timeout 300 python scratch/ode_plan_time.py 90 50 >> $O/ode_plans.txt 2>&1; tail -4 $O/ode_plans.txt
timeout 900 python -m pytest tests/test_gpu_weight_seeds.py tests/test_gpu_bench.py -q -x -s -p no:cacheprovider > $O/new_tests_b.log 2>&1; tail -15 $O/new_tests_b.log
GP_PROXY_REPORT=$O/accuracy_proxy.txt GP_PROXY_INSTANCES=512 timeout 1200 python -m pytest tests/test_gpu_trained_regime.py -q -x -s -p no:cacheprovider > $O/trained.log 2>&1; tail -30 $O/trained.log
if [ "$1" != "notests" ]; then
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1; tail -25 $O/pytest_gpu.log
fi
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err; cut -c1-600 $O/bench_line.json

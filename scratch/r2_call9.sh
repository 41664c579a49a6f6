#!/bin/bash
O=gpurun_out/r2c9; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -14 $O/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-secondary --sampler ode > $O/bench_ode.json 2>$O/bench.err; tail -c 900 $O/bench_ode.json
python scratch/bench_tracking.py 16 64 > $O/tracking.txt 2>&1; tail -5 $O/tracking.txt

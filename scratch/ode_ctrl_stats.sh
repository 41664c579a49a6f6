#!/bin/bash
# per-kernel time of the RK45 controller kernels under the ODE bench (tuning): bash scratch/ode_ctrl_stats.sh
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ode
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ode -- python /root/repo/bench.py --no-cpu-baseline --no-secondary --sampler ode > /tmp/ode.log 2>&1
tail -1 /tmp/ode.log | cut -c1-200
f=$(find /tmp/prof_ode -name "*kernel_stats.csv" | head -1)
grep -E "decide|reset|slot0|embed" $f | cut -c1-200

#!/bin/bash
# per-kernel time under the tracking bench (tuning): bash scratch/trk_stats.sh [sequences]
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_trk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trk -- python /root/repo/scratch/bench_tracking.py ${1:-64} > /tmp/trk.log 2>&1
tail -2 /tmp/trk.log
f=$(find /tmp/prof_trk -name "*kernel_stats.csv" | head -1)
head -16 $f | cut -c1-150 | awk -F'","' '{printf "%-110s %8s %12s %10s\n", substr($1,2,108), $2, $3, $4}'

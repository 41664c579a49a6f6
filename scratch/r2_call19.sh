#!/bin/bash
cd $GRAFT_REPO_ROOT
for g in 5 10 20; do timeout 300 python bench.py --no-cpu-baseline --no-secondary --batches-per-launch $g --steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('G=$g', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'], d['timing'])"; done

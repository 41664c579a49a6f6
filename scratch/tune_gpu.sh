#!/bin/bash
GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip_timing.so python scratch/timing.py 64 50 | grep "wave 0\|wave 3"
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-pipeline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('seq', d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'])"
python bench.py --steps 30 --warmup 4 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipe', d['value'], d['ms_per_step'], d['roofline'].get('in_situ_avg_launch_us'))"

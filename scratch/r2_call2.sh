#!/bin/bash
# round 2, GPU call 2: re-run of the adjusted tests + measurements that decide the perf work
set -x
O=gpurun_out/r2c2; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_energy_sampling.py tests/test_gpu_sampler.py -m gpu -q -rP > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
# phase anatomy of the sampler kernels (tuning build)
GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip_timing.so timeout 300 python scratch/timing.py 320 50 > $O/timing_320.txt 2>&1
GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip_timing.so timeout 300 python scratch/timing.py 64 50 > $O/timing_64.txt 2>&1
# FPS on the side stream: on / off
timeout 300 python bench.py --no-cpu-baseline --no-secondary > $O/bench_fps_ahead.json 2>$O/bench1.err
timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-fps-ahead > $O/bench_no_fps_ahead.json 2>$O/bench2.err
timeout 300 python bench.py --no-cpu-baseline --no-secondary --overlap > $O/bench_overlap.json 2>$O/bench3.err
# encoder alone
timeout 300 python scratch/enc_profile.py 320 > $O/enc_320.txt 2>&1
timeout 300 python scratch/enc_profile.py 64 > $O/enc_64.txt 2>&1
timeout 300 python scratch/enc_profile.py 256 > $O/enc_256.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_enc320 -- python $GRAFT_REPO_ROOT/scratch/enc_profile.py 320 20 > $GRAFT_REPO_ROOT/$O/prof_enc320.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_enc320 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/enc320_kernel_stats.csv
rm -rf $O/prof_enc320
tail -3 $O/pytest.log; cat $O/timing_320.txt | tail -9; cat $O/enc_*.txt

#!/bin/bash
# round 2: one GPU session that regenerates everything under profiles/r2_* (run through gpurun from the repo root)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout 200 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench.err; cut -c1-300 $O/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-secondary > $O/bench_line_under_rocprof.json 2> $O/prof_bench.err
cp /tmp/prof_bench/*/bench_kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null || find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
timeout 300 python bench.py --no-cpu-baseline --no-secondary --overlap > $O/bench_line_overlap.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-secondary --batches-per-launch 1 > $O/bench_line_g1.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-secondary --sampler ode > $O/bench_line_ode.json 2>/dev/null
timeout 300 python bench.py --no-cpu-baseline --no-secondary --pipeline full --batch 256 > $O/bench_line_full256.json 2>/dev/null
GP_BENCH_ONE_DEVICE=1 timeout 400 python bench.py --gpus 2 --no-cpu-baseline --no-secondary > $O/bench_line_2ranks_one_device.json 2>$O/bench_2ranks.err
timeout 300 python scratch/bench_tracking.py 16 64 128 > $O/tracking.txt 2>/dev/null
for B in 64 320 640; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_fetch_$B -o run -- python scratch/microbench.py $B 50 > $O/pmc_fetch_$B.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_write_$B -o run -- python scratch/microbench.py $B 50 > $O/pmc_write_$B.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d /tmp/pmc_sq_640 -o run -- python scratch/microbench.py 640 50 > $O/pmc_sq_640.log 2>&1
F64=$(find /tmp/pmc_fetch_64 -name "*.db" | head -1); W64=$(find /tmp/pmc_write_64 -name "*.db" | head -1)
F320=$(find /tmp/pmc_fetch_320 -name "*.db" | head -1); W320=$(find /tmp/pmc_write_320 -name "*.db" | head -1)
F640=$(find /tmp/pmc_fetch_640 -name "*.db" | head -1); W640=$(find /tmp/pmc_write_640 -name "*.db" | head -1)
python scratch/pmc_traffic.py $O/pmc_traffic.json 64:$F64:$W64 320:$F320:$W320 640:$F640:$W640 > $O/pmc_traffic.log 2>&1; tail -12 $O/pmc_traffic.log
python scratch/pmc_summary.py $(find /tmp/pmc_sq_640 -name "*.db" | head -1) > $O/pmc_sq_summary.txt 2>&1; head -5 $O/pmc_sq_summary.txt
ls $O

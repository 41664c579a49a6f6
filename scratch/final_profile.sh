#!/bin/bash
# one GPU session that regenerates everything under profiles/ (run through gpurun from the repo root)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout 200 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py > $O/bench_line.json 2> $O/bench.err; cut -c1-300 $O/bench_line.json
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline > $O/bench_line_under_rocprof.json 2> $O/prof_bench.err
timeout 200 rocprofv3 --kernel-trace -d $O/prof_seq -o run -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline > /dev/null 2> $O/prof_seq.err
python scratch/prof_summary.py $O/prof_seq/run_results.db > $O/kernel_summary.txt 2>&1
timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --overlap > $O/bench_line_overlap.json 2>/dev/null
timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --batches-per-launch 1 --overlap > $O/bench_line_g1_overlap.json 2>/dev/null
timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --batches-per-launch 1 > $O/bench_line_g1.json 2>/dev/null
timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --sampler ode > $O/bench_line_ode.json 2>/dev/null
timeout 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --sampler ode --batches-per-launch 1 > $O/bench_line_ode_g1.json 2>/dev/null
timeout 300 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --pipeline full --batch 256 > $O/bench_line_full256.json 2>/dev/null
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --pipeline full --batch 256 --sampler ode > $O/bench_line_full256_ode.json 2>/dev/null
timeout 300 python scratch/bench_tracking.py 16 32 64 128 > $O/tracking.txt 2>/dev/null
timeout 120 python scratch/bench_pre.py 6 > $O/preprocess.txt 2>/dev/null; timeout 120 python scratch/bench_pre.py 64 >> $O/preprocess.txt 2>/dev/null
for B in 64 320; do
  timeout 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch_$B -o run -- python scratch/microbench.py $B 50 > $O/pmc_fetch_$B.log 2>&1
  timeout 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write_$B -o run -- python scratch/microbench.py $B 50 > $O/pmc_write_$B.log 2>&1
done
timeout 150 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d $O/pmc_sq_320 -o run -- python scratch/microbench.py 320 50 > $O/pmc_sq_320.log 2>&1
python scratch/pmc_traffic.py $O/pmc_traffic.json 64:$O/pmc_fetch_64/run_results.db:$O/pmc_write_64/run_results.db 320:$O/pmc_fetch_320/run_results.db:$O/pmc_write_320/run_results.db > $O/pmc_traffic.log 2>&1; tail -20 $O/pmc_traffic.log
python scratch/pmc_summary.py $O/pmc_sq_320/run_results.db > $O/pmc_sq_summary.txt 2>&1
find $O -name "*.db" -size +30M -delete
ls $O $O/prof_bench

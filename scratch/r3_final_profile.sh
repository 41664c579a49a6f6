#!/bin/bash
# round 3: one GPU session that regenerates everything under profiles/r3_* (run through gpurun from the repo root)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
timeout 200 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err; cut -c1-400 $O/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-secondary > $O/bench_line_under_rocprof.json 2> $O/prof_bench.err
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
{
for v in "--overlap" "--batches-per-launch 1" "--batches-per-launch 5" "--batches-per-launch 20" "--sampler ode" "--pipeline full --batch 256" "--tracking --sequences 64 --steps 10 --warmup 3" "--tracking --sequences 1 --steps 20 --warmup 3"; do
  echo "== bench.py --no-cpu-baseline --no-secondary $v"; timeout 300 python bench.py --no-cpu-baseline --no-secondary $v 2>/dev/null
done
echo "== GP_BENCH_ONE_DEVICE=1 bench.py --gpus 2 --no-cpu-baseline --no-secondary"; GP_BENCH_ONE_DEVICE=1 timeout 400 python bench.py --gpus 2 --no-cpu-baseline --no-secondary 2>$O/bench_2ranks.err
} > $O/bench_variants.txt
timeout 300 python scratch/bench_tracking.py 16 64 128 > $O/tracking.txt 2>/dev/null
timeout 200 python scratch/chain_check.py > $O/plans.txt 2>&1
./scratch/occ/mfma_power > $O/mfma_power.txt 2>&1
bash scratch/enc_kernel_stats.sh 320 $O/encoder320_kernel_stats.txt > /dev/null 2>&1
for B in 64 640; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_fetch_$B -o run -- python scratch/microbench.py $B 50 > $O/pmc_fetch_$B.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_write_$B -o run -- python scratch/microbench.py $B 50 > $O/pmc_write_$B.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d /tmp/pmc_sq_640 -o run -- python scratch/microbench.py 640 50 > $O/pmc_sq_640.log 2>&1
F64=$(find /tmp/pmc_fetch_64 -name "*.db" | head -1); W64=$(find /tmp/pmc_write_64 -name "*.db" | head -1)
F640=$(find /tmp/pmc_fetch_640 -name "*.db" | head -1); W640=$(find /tmp/pmc_write_640 -name "*.db" | head -1)
python scratch/pmc_traffic.py $O/pmc_traffic.json 64:$F64:$W64 640:$F640:$W640 > $O/pmc_traffic.log 2>&1; tail -12 $O/pmc_traffic.log
python scratch/pmc_summary.py $(find /tmp/pmc_sq_640 -name "*.db" | head -1) > $O/pmc_sq_summary.txt 2>&1; head -12 $O/pmc_sq_summary.txt
ls $O

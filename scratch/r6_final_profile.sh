#!/bin/bash
# round 6: one GPU session that regenerates everything under profiles/r6_* (run through gpurun from the repo root)
#   bash scratch/r6_final_profile.sh [notests]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6final; rm -rf $O; mkdir -p $O
if [ "$1" != "notests" ]; then
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1; tail -22 $O/pytest_gpu.log
timeout 200 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err; cut -c1-400 $O/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o bench -- python bench.py --no-cpu-baseline --no-secondary > $O/bench_line_under_rocprof.json 2> $O/prof_bench.err
find /tmp/prof_bench -name "*kernel_stats.csv" -exec cp {} $O/bench_kernel_stats.csv \;
{
for v in "--batches-per-launch 1" "--batches-per-launch 5" "--batches-per-launch 20" "--sampler ode" "--pipeline full --batch 256" "--tracking --sequences 64 --steps 10 --warmup 3" "--tracking --sequences 1 --steps 20 --warmup 8"; do
  echo "== bench.py --no-cpu-baseline --no-secondary $v"; timeout 300 python bench.py --no-cpu-baseline $( [[ "$v" == *"sequences 1 "* ]] || echo --no-secondary ) $v 2>/dev/null
done
echo "== GP_BENCH_ONE_DEVICE=1 bench.py --gpus 8 --no-cpu-baseline --no-secondary (eight ranks sharing the one device, collectives on gloo)"; GP_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 8 --no-cpu-baseline --no-secondary 2>$O/bench_8ranks.err
} > $O/bench_variants.txt
timeout 300 python scratch/bench_tracking.py 16 64 128 > $O/tracking.txt 2>/dev/null
timeout 300 python scratch/headsplit_plans.py > $O/plans.txt 2>&1
{ echo; echo "== RK45 launch plans at the eval_single shape (scratch/ode_plan_time.py)"; timeout 300 python scratch/ode_plan_time.py 2>&1; timeout 200 python scratch/ode_plan_time.py 90 50 2>&1; } >> $O/plans.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dropin -o dropin -- python bench.py --only-drop-in > $O/drop_in_under_rocprof.json 2> $O/prof_dropin.err
find /tmp/prof_dropin -name "*kernel_stats.csv" -exec cp {} $O/drop_in_kernel_stats.csv \;
for mode in forward graph; do for B in 5 64 320 640; do timeout 100 python scratch/enc_profile.py $B 30 $mode 2>/dev/null | tail -1; done; done > $O/encoder_wall.txt
bash scratch/enc_kernel_stats.sh 320 $O/encoder320_kernel_stats.txt > /dev/null 2>&1
bash scratch/enc_timeline.sh 320 graph $O/encoder320_timeline.txt > /dev/null 2>&1
timeout 120 python scratch/fps_waves.py > $O/fps_one_wave.txt 2>&1
timeout 120 python scratch/ubench/run.py > $O/ubench_valu_beside_mfma.txt 2>&1
rm -rf /tmp/prof_trk; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trk -o trk -- python scratch/track_one.py > $O/track_one.log 2>&1
python - "$(find /tmp/prof_trk -name '*kernel_stats.csv' | head -1)" > $O/tracking_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    per_frame = float(r["TotalDurationNs"]) / 106 / 1e3
    tot += per_frame
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
    if per_frame >= 10: print(f"{name:66s} calls/frame {int(r['Calls']) / 106:6.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  per frame {per_frame:8.1f} us")
print(f"kernel time per frame {tot:.1f} us (sum over all streams; 106 frames)")
PY
grep '^frames' $O/track_one.log | tail -1 >> $O/tracking_kernels.txt
for B in 64 640; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_fetch_$B -o run -- python scratch/microbench.py $B 50 > $O/pmc_fetch_$B.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_write_$B -o run -- python scratch/microbench.py $B 50 > $O/pmc_write_$B.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d /tmp/pmc_sq_640 -o run -- python scratch/microbench.py 640 50 > $O/pmc_sq_640.log 2>&1
F64=$(find /tmp/pmc_fetch_64 -name "*.db" | head -1); W64=$(find /tmp/pmc_write_64 -name "*.db" | head -1)
F640=$(find /tmp/pmc_fetch_640 -name "*.db" | head -1); W640=$(find /tmp/pmc_write_640 -name "*.db" | head -1)
python scratch/pmc_traffic.py $O/pmc_traffic.json 64:$F64:$W64 640:$F640:$W640 > $O/pmc_traffic.log 2>&1; tail -12 $O/pmc_traffic.log
python scratch/pmc_summary.py $(find /tmp/pmc_sq_640 -name "*.db" | head -1) > $O/pmc_sq_summary.txt 2>&1
# the RK45 attempt at the eval_single shape under its three plans: SQ counters of rk45_attempt_shared_kernel<48> against the stage kernels
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace -d /tmp/pmc_sq_ode -o run -- python scratch/ode_plan_time.py > $O/pmc_sq_ode.log 2>&1
{ echo "---- RK45 at 256 clouds x 50 candidates (scratch/ode_plan_time.py): attempt / stage kernels of the three plans"; python scratch/pmc_summary.py $(find /tmp/pmc_sq_ode -name "*.db" | head -1) rk45_; } >> $O/pmc_sq_summary.txt 2>&1
timeout 300 python scratch/shared_plan_soak.py > $O/shared_plan_soak.txt 2>&1; timeout 200 python scratch/shared_plan_control.py >> $O/shared_plan_soak.txt 2>&1; tail -3 $O/shared_plan_soak.txt
ls $O

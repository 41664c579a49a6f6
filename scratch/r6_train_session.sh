#!/bin/bash
# round 6: trains the two networks on synthetic posed clouds (<= 15 GPU-minutes) and profiles the training step
#   bash scratch/r6_train_session.sh [minutes_score] [minutes_energy]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6train; rm -rf $O; mkdir -p $O
MS=${1:-8}; ME=${2:-5}
# 1. every phase once, small (fails fast if anything is wrong on the device)
timeout 300 python scratch/train_synth.py --steps 12 --clouds 2048 --out /tmp/probe > $O/probe.log 2>&1 || { tail -30 $O/probe.log; exit 1; }
grep "steps of" $O/probe.log
# 2. the kernels of a training step: rocprofv3 over 12 steps of each phase
rm -rf /tmp/prof_train; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o train -- python scratch/train_synth.py --steps 12 --clouds 2048 --out /tmp/probe2 > $O/prof_run.log 2>&1
find /tmp/prof_train -name "*kernel_stats.csv" -exec cp {} $O/training_kernel_stats.csv \;
python scratch/training_kernel_split.py $O/training_kernel_stats.csv > $O/training_step.txt 2>&1
grep "steps of" $O/prof_run.log >> $O/training_step.txt; cat $O/training_step.txt
# 3. the bounded training run
timeout $(( (MS + ME) * 60 + 400 )) python scratch/train_synth.py --minutes-score $MS --minutes-energy $ME --out $O > $O/train.log 2>&1
tail -25 $O/train.log; ls -la $O

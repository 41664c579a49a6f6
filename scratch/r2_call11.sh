#!/bin/bash
O=gpurun_out/r2c11; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in "" _diet; do
  for fl in "" "--overlap"; do
    GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip$v.so timeout 300 python bench.py --no-cpu-baseline --no-secondary $fl > $O/b.json 2>/dev/null
    python - "$v" "$fl" <<'PY' >> $O/out.txt
import json,sys
d=json.load(open("gpurun_out/r2c11/b.json"))
print(f"lib{sys.argv[1] or '_default'} {sys.argv[2] or 'one-stream'}: {d['value']} poses/s, {d['ms_per_step']} ms/step, sampler launch alone {d['roofline']['avg_launch_us']} us, in situ {d['roofline'].get('in_situ_avg_launch_us')} us, one-batch {d['one_batch_per_launch']['value']}")
PY
  done
done
GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip_diet.so timeout 600 python -m pytest tests/test_gpu_score.py tests/test_gpu_tile32.py tests/test_gpu_sampler.py -m gpu -q > $O/pytest_diet.log 2>&1; tail -2 $O/pytest_diet.log
timeout 300 python -m pytest tests/test_training.py -m gpu -q > $O/pytest_train.log 2>&1; tail -2 $O/pytest_train.log
cat $O/out.txt

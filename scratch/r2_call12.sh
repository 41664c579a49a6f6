#!/bin/bash
O=gpurun_out/r2c12; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in "" _diet16; do
  GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip$v.so timeout 200 python scratch/pc_time.py 64 128 320 2>&1 | grep -v amdgpu.ids >> $O/out.txt
done
cat $O/out.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip_diet16.so timeout 600 python -m pytest tests/test_gpu_score.py tests/test_gpu_tile32.py tests/test_gpu_sampler.py tests/test_gpu_pipeline.py -m gpu -q > $O/pytest_diet16.log 2>&1; tail -2 $O/pytest_diet16.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2>/dev/null; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2c12/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d["one_batch_per_launch"]["value"], d["one_batch_per_launch"]["frac"], d["ode_100"]["value"], d["full_pipeline_256"]["value"])
PY

#!/bin/bash
O=gpurun_out/r2c8; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py tests/test_gpu_sampler.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 400 python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/r2c8/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["one_batch_per_launch"], d["ode_100"], d["full_pipeline_256"])
PY
timeout 300 python bench.py --no-cpu-baseline --no-secondary --pipeline full --batch 256 > $O/bench_full.json 2>>$O/bench.err; tail -c 600 $O/bench_full.json

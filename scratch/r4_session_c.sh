#!/bin/bash
# round 4, session C: FPS packed distance update, ball query two centres in flight, tracking frame with the energy encoder under the solve,
# configs[0] test + legs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_ops.py tests/test_gpu_encoder.py tests/test_gpu_pipeline.py tests/test_gpu_edge_cases.py \
   "tests/test_gpu_fullsize.py::test_config0_single_object" tests/test_gpu_fullsize.py::test_encoder_vs_oracle_at_bench_sizes > $O/pytest.log 2>&1; tail -5 $O/pytest.log
{
for B in 5 64 320; do
  for mode in forward graph; do timeout 100 python scratch/enc_profile.py $B 30 $mode 2>/dev/null | tail -1; done
done
} > $O/enc_wall.txt; cat $O/enc_wall.txt
bash scratch/enc_kernel_stats.sh 320 $O/encoder320_kernel_stats.txt > /dev/null 2>&1; grep "fps\|ball\|sum" $O/encoder320_kernel_stats.txt
timeout 300 python scratch/bench_tracking.py 16 64 > $O/tracking.txt 2>/dev/null; cat $O/tracking.txt
timeout 500 python bench.py > $O/bench_line.json 2> $O/bench.err; python - <<'PY'
import json
l = json.loads(open("gpurun_out/r4c/bench_line.json").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value", "ms_per_step")}, l["roofline"]["frac"], l["roofline"]["avg_launch_us"])
for k in ("one_batch_per_launch", "ode_100", "full_pipeline_256", "drop_in_eval_single", "config0_single_object", "cpu_baseline"):
    print(k, json.dumps(l.get(k))[:600])
PY
tail -3 $O/bench.err

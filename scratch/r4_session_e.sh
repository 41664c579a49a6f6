#!/bin/bash
# round 4, session E: layer-1 fmaf chain + three workgroups per CU on the level-1 kernel (A/B against the base library), tracking frame timeline,
# energy-model leg, likelihood chain test
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4e; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest -q -m gpu -p no:cacheprovider tests/test_gpu_encoder.py tests/test_gpu_sa_paths.py tests/test_gpu_chain_vjp.py::test_likelihood_chain_vs_tile \
   tests/test_gpu_fullsize.py::test_encoder_vs_oracle_at_bench_sizes > $O/pytest.log 2>&1; tail -4 $O/pytest.log
{
for B in 64 320; do
  GENPOSE_HIP_LIB=$PWD/genpose_amd/lib/libgenpose_hip_base.so timeout 100 python scratch/enc_profile.py $B 30 graph 2>/dev/null | tail -1 | sed 's/^/base /'
  timeout 100 python scratch/enc_profile.py $B 30 graph 2>/dev/null | tail -1
done
} > $O/enc_wall.txt; cat $O/enc_wall.txt
bash scratch/enc_kernel_stats.sh 320 $O/encoder320_kernel_stats.txt > /dev/null 2>&1; cat $O/encoder320_kernel_stats.txt
rm -rf /tmp/prof_trk; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trk -o trk -- python scratch/track_one.py > $O/track_one.log 2>&1; tail -1 $O/track_one.log
F=$(find /tmp/prof_trk -name "*kernel_stats.csv" | head -1); python - "$F" > $O/tracking_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0
for r in rows:
    per_frame = float(r["TotalDurationNs"]) / 106 / 1e3
    tot += per_frame
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
    print(f"{name:66s} calls/frame {int(r['Calls']) / 106:6.1f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  per frame {per_frame:8.1f} us")
print(f"kernel time per frame {tot:.1f} us (sum over all streams)")
PY
head -24 $O/tracking_kernels.txt; tail -1 $O/tracking_kernels.txt
timeout 100 python scratch/track_one.py 2>/dev/null | tail -1

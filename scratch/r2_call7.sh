#!/bin/bash
O=gpurun_out/r2c7; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_sa_paths.py tests/test_gpu_ops.py -m gpu -q -x > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
timeout 200 python scratch/enc_profile.py 320 > $O/enc.txt 2>&1; timeout 200 python scratch/enc_profile.py 64 >> $O/enc.txt 2>&1; grep encoder $O/enc.txt
cd /tmp && export TMPDIR=/tmp
for B in 320; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc$B -- python $R/scratch/enc_profile.py $B 20 > $R/$O/prof_enc$B.log 2>&1
  f=$(find /tmp/prof_enc$B -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/enc${B}_kernel_stats.csv
done
cd $R
python - <<'PY'
import csv
for B in (320,):
    rows = list(csv.DictReader(open(f"gpurun_out/r2c7/enc{B}_kernel_stats.csv")))
    print(f"== encoder B={B}: kernel, calls, avg us, total ms")
    for r in rows[:12]:
        print(f"{r['Name'][:90]:<92}{r['Calls']:>5}{float(r['AverageNs'])/1e3:>10.1f}{float(r['TotalDurationNs'])/1e6:>9.2f}")
PY

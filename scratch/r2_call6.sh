#!/bin/bash
O=gpurun_out/r2c6; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for B in 320 64; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc$B -- python $R/scratch/enc_profile.py $B 20 > $R/$O/prof_enc$B.log 2>&1
  f=$(find /tmp/prof_enc$B -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/enc${B}_kernel_stats.csv
done
cd $R
python - <<'PY'
import csv
for B in (320, 64):
    rows = list(csv.DictReader(open(f"gpurun_out/r2c6/enc{B}_kernel_stats.csv")))
    print(f"== encoder B={B}: kernel, calls, avg us, total ms")
    for r in rows[:16]:
        print(f"{r['Name'][:90]:<92}{r['Calls']:>5}{float(r['AverageNs'])/1e3:>10.1f}{float(r['TotalDurationNs'])/1e6:>9.2f}")
PY

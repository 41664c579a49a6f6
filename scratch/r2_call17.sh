#!/bin/bash
O=gpurun_out/r2c17; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd $R; timeout 200 python scratch/enc_profile.py 5 50 2>&1 | grep encoder
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc5 -- python $R/scratch/enc_profile.py 5 50 > $R/$O/prof.log 2>&1
f=$(find /tmp/prof_enc5 -name "*kernel_stats.csv" | head -1); cp "$f" $R/$O/enc5_kernel_stats.csv
cd $R; python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r2c17/enc5_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("sum of kernel time per pass (53 passes): %.1f us, launches per pass %.1f"%(tot/53/1e3, calls/53))
for r in rows[:8]: print(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3)
PY

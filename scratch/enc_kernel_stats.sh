#!/bin/bash
# per-kernel times of the encoder alone: bash scratch/enc_kernel_stats.sh <clouds> <outfile>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=${1:-320}; OUT=${2:-gpurun_out/enc_stats.txt}
rm -rf /tmp/prof_enc; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_enc -o enc -- python scratch/enc_profile.py $B 20 > /tmp/enc_run.log 2>&1
F=$(find /tmp/prof_enc -name "*kernel_stats.csv" | head -1)
python - "$F" "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], "w")
tot = 0
for r in rows:
    name = r["Name"]
    if "at::native" in name or "rocclr" in name: continue
    calls, total = int(r["Calls"]), float(r["TotalDurationNs"])
    per_pass = total / 23 / 1e3  # 3 warm-up + 20 timed passes
    tot += per_pass
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
    line = f"{short:72s} calls {calls:5d}  avg {float(r['AverageNs'])/1e3:9.1f} us  per pass {per_pass:9.1f} us"
    print(line); out.write(line + "\n")
print(f"sum per pass {tot:.1f} us"); out.write(f"sum per pass {tot:.1f} us\n")
PY
tail -1 /tmp/enc_run.log

#!/bin/bash
# timeline of ONE encoder pass (the last one of the run): bash scratch/enc_timeline.sh <clouds> <mode> <outfile> [precision]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B=${1:-320}; MODE=${2:-graph}; OUT=${3:-gpurun_out/enc_timeline.txt}; PREC=${4:-f32}
rm -rf /tmp/prof_tl; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tl -o tl -- python scratch/enc_profile.py $B 6 $MODE $PREC > /tmp/tl_run.log 2>&1
F=$(find /tmp/prof_tl -name "*kernel_trace.csv" | head -1)
python - "$F" "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "at::native" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass starts at the last fps_chain launch that samples level 0 (first fps kernel of a pass)
starts = [i for i, r in enumerate(rows) if "fps_" in r["Kernel_Name"]]
# passes launch one or two fps chains; take the last chain whose predecessor is not an fps chain within 50 us
first = starts[-1]
for i in reversed(starts):
    if int(rows[first]["Start_Timestamp"]) - int(rows[i]["Start_Timestamp"]) < 400_000: first = i
t0 = int(rows[first]["Start_Timestamp"])
out = open(sys.argv[2], "w")
end = 0
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    gap = s - end if end else 0
    line = f"{s/1e3:9.1f} -> {e/1e3:9.1f} us  ({(e-s)/1e3:8.1f} us, {'+' if gap >= 0 else ''}{gap/1e3:7.1f} after the latest end)  q{r.get('Queue_Id','?')}  {name}"
    print(line); out.write(line + "\n")
    end = max(end, e)
PY
tail -1 /tmp/tl_run.log

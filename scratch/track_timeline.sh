#!/bin/bash
# timeline of ONE tracking frame (a steady-state frame near the end of the run): bash scratch/track_timeline.sh <outfile>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/track_timeline.txt}
rm -rf /tmp/prof_ttl; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ttl -o tl -- python scratch/track_one.py > /tmp/ttl_run.log 2>&1
F=$(find /tmp/prof_ttl -name "*kernel_trace.csv" | head -1)
python - "$F" "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a frame starts with the copy of the clouds into graph A's input followed by the level-0 sampling launch; frames = groups between sampling launches
# that are more than 600 us apart
fps = [i for i, r in enumerate(rows) if "fps_chain" in r["Kernel_Name"]]
starts = [fps[0]]
for i in fps[1:]:
    if int(rows[i]["Start_Timestamp"]) - int(rows[starts[-1]]["Start_Timestamp"]) > 900_000: starts.append(i)
out = open(sys.argv[2], "w")
def emit(line):
    print(line); out.write(line + "\n")
for which in (-12, -5):   # two steady-state frames
    a, b = starts[which], starts[which + 1]
    t0 = int(rows[a]["Start_Timestamp"])
    emit(f"=== frame {len(starts) + which}: {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us from its level-0 sampling launch to the next frame's")
    end = {}
    busy = 0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        q = r.get("Queue_Id", "?")
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:52]
        gap = s - end.get(q, s)
        emit(f"{s/1e3:8.1f} -> {e/1e3:8.1f} us ({(e-s)/1e3:6.1f}, {gap/1e3:+7.1f} after the queue's last end)  q{q}  {name}")
        end[q] = e
PY
tail -1 /tmp/ttl_run.log

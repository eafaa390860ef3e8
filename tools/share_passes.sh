#!/bin/bash
# tools/share_passes.sh N: per-launch durations of one rank's share of the C2 frame at N ranks (kernel trace of tools/rank_share.py)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-8}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/share_tr
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/share_tr -- python3 $R/tools/rank_share.py $N || exit 1
python3 - <<'PY'
import csv, glob, re
rows = []
for f in glob.glob("/tmp/share_tr/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [r for r in rows if re.search(r"(intersect|shade|resolve)_kernel", r["Kernel_Name"])]
res = [i for i, r in enumerate(last) if "resolve" in r["Kernel_Name"]]
last = last[res[-2] + 1:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    print(f"{m.group(1):18s} start {(int(r['Start_Timestamp']) - t0) / 1e6:7.3f} ms  duration {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:6.3f} ms")
PY

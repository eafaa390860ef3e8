#!/bin/bash
# tools/ktrace.sh <tag> [script]: rocprofv3 kernel trace of a short script (default tools/pass_times.py, 13 C2 frames) and
# the per-launch durations of its last frame -> gpurun_out/<tag>_passes.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-ktrace}
SCRIPT=${2:-$R/tools/pass_times.py}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $SCRIPT > $OUT/run.log 2>&1 || { echo "trace run failed"; tail -5 $OUT/run.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [r for r in rows if re.search(r"(intersect|shade|resolve|exceptions)_kernel", r["Kernel_Name"])]
n_res = [i for i, r in enumerate(last) if "resolve" in r["Kernel_Name"]]
if len(n_res) >= 2:
    last = last[n_res[-2] + 1:]
tot = {}
with open(out + "_passes.txt", "w") as fh:
    for r in last:
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
        ms = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        name = m.group(1) + (m.group(2) or '')
        tot[m.group(1)] = tot.get(m.group(1), 0) + ms
        fh.write(f"{name} {ms:.3f}\n")
    fh.write("totals " + " ".join(f"{k}={v:.3f}" for k, v in tot.items()) + "\n")
print(open(out + "_passes.txt").read())
PY
rm -rf $OUT/trace
tail -2 $OUT/run.log

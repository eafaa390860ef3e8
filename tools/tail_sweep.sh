#!/bin/bash
# tools/tail_sweep.sh: per-launch durations of a C2 frame (tools/ktrace.sh) over the intersect scheduling knobs, to see
# how the small late passes respond:   prints, per setting, the intersect passes 0..8 in ms
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export RT_BUILD_KEY=0x12345678
run() {
    echo -n "$1: "
    env $1 bash $R/tools/ktrace.sh tail_sweep > /dev/null 2>&1 || { echo failed; exit 1; }
    grep "^intersect" $R/gpurun_out/tail_sweep_passes.txt | awk '{printf "%s ", $NF}'; grep "^totals" $R/gpurun_out/tail_sweep_passes.txt
}
run RT_X=0
for s in RT_QUORUM=1 RT_QUORUM=4 RT_QUORUM=16 RT_PRIM_QUORUM=1 RT_PRIM_QUORUM=4 RT_PRIM_QUORUM=32 RT_FINISH_QUORUM=4 RT_FINISH_QUORUM=16 RT_MIN_CHUNK=16 RT_MIN_CHUNK=32 RT_MIN_CHUNK=128; do run $s; done

#!/bin/bash
# tools/c2_ta.sh <tag>: texture-address and vector-cache counters of every kernel of the 1080p, 64 spp frame (one rocprofv3
# --pmc pass of tools/pass_times.py), to tell kernels bound by address processing from kernels bound by arithmetic
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-c2ta}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/p1 -- python3 $R/tools/pass_times.py > $OUT/p1.log 2>&1 || { echo "pmc pass failed"; tail -3 $OUT/p1.log; exit 1; }
python3 $R/tools/pmc_summary.py $OUT/p1 > $OUT/summary.txt
rm -rf $OUT/p1
grep -A10 -E "^(shade|intersect|resolve)" $OUT/summary.txt

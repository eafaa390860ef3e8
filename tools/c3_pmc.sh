#!/bin/bash
# tools/c3_pmc.sh <tag> [library]: vector-cache counters of tools/c3_quick.py (the 100k-primitive scene read from memory)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-c3pmc}
[ -n "$2" ] && export RT_ENGINE_LIB=$2
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_TA_BUSY_sum SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/p1 -- python3 $R/tools/c3_quick.py > $OUT/p1.log 2>&1 || { echo "pmc pass failed"; tail -3 $OUT/p1.log; exit 1; }
python3 $R/tools/pmc_summary.py $OUT/p1 > $OUT/summary.txt
rm -rf $OUT/p1
grep -A12 "^intersect_kernel<false" $OUT/summary.txt

#!/bin/bash
# tools/c3_lds_pmc.sh <tag> [library]: LDS and issue counters of tools/c3_quick.py (the 100k-primitive scene read from memory)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-c3lds}
[ -n "$2" ] && export RT_ENGINE_LIB=$2
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -- python3 $R/tools/c3_quick.py > $OUT/p1.log 2>&1 || { echo "pmc pass 1 failed"; tail -3 $OUT/p1.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 $R/tools/c3_quick.py > $OUT/p2.log 2>&1 || { echo "pmc pass 2 failed"; tail -3 $OUT/p2.log; exit 1; }
python3 $R/tools/pmc_summary.py $OUT/p1 $OUT/p2 > $OUT/summary.txt
rm -rf $OUT/p1 $OUT/p2
grep -A18 "^intersect_kernel<false" $OUT/summary.txt

#!/bin/bash
# tools/icache.sh: instruction-cache counters of one bench.py frame (own rocprofv3 --pmc run)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ic
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/ic -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > /tmp/ic.log 2>&1 || { tail -3 /tmp/ic.log; exit 1; }
python3 $R/tools/pmc_summary.py /tmp/ic | grep -A9 -E "^(intersect|shade)"

#!/bin/bash
# tools/sweep.sh: bench.py (C2 only) over the intersect scheduling knobs of the bounce passes; prints ms per frame
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { env "$@" timeout -k 10 120 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', round(d['per_rank']['intersect_kernel_ms'][0],2), 'ms in intersect')"; }
echo -n "defaults: "; run RT_X=0 || exit 1
for rb in 8 12 24 32; do echo -n "refill_at_bounce=$rb: "; run RT_REFILL_AT_BOUNCE=$rb || exit 1; done
for qu in 4 12 16 24; do echo -n "quorum=$qu: "; run RT_QUORUM=$qu || exit 1; done
for pq in 8 24 32; do echo -n "prim_quorum=$pq: "; run RT_PRIM_QUORUM=$pq || exit 1; done
for fq in 4 8; do echo -n "finish_quorum=$fq: "; run RT_FINISH_QUORUM=$fq || exit 1; done
for mc in 32 128; do echo -n "min_chunk=$mc: "; run RT_MIN_CHUNK=$mc || exit 1; done
for g in 8 32; do echo -n "guide=$g: "; run RT_GUIDE=$g || exit 1; done
echo -n "defaults: "; run RT_X=0 || exit 1

#!/bin/bash
# tools/c3_sweep.sh: tools/c3_quick.py (100k-primitive scene read from memory) over the intersect scheduling knobs
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { env "$@" timeout -k 10 120 python3 $R/tools/c3_quick.py 2>/dev/null | tail -1 | cut -d, -f1-2; }
echo -n "defaults: "; run RT_X=0 || exit 1
for rb in 4 8 24 32 48; do echo -n "refill_at_bounce=$rb: "; run RT_REFILL_AT_BOUNCE=$rb || exit 1; done
for qu in 1 4 16 24 32; do echo -n "quorum=$qu: "; run RT_QUORUM=$qu || exit 1; done
for pq in 4 8 32 48; do echo -n "prim_quorum=$pq: "; run RT_PRIM_QUORUM=$pq || exit 1; done
for mc in 32 128; do echo -n "min_chunk=$mc: "; run RT_MIN_CHUNK=$mc || exit 1; done
echo -n "defaults: "; run RT_X=0 || exit 1

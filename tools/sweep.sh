#!/bin/bash
# tools/sweep.sh: bench.py over the intersect scheduling knobs (RT_REFILL_AT, RT_QUORUM, RT_PRIM_QUORUM, RT_CHUNK); prints ms per frame
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() { env "$@" timeout -k 10 120 python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', round(d['value']), 'Mrays/s')"; }
for ra in 24 32 40; do for qu in 4 8 16; do for pq in 4 8 16; do echo -n "refill_at=$ra quorum=$qu prim_quorum=$pq: "; run RT_REFILL_AT=$ra RT_QUORUM=$qu RT_PRIM_QUORUM=$pq || exit 1; done; done; done
for ch in 256 512 1024; do echo -n "chunk=$ch: "; run RT_CHUNK=$ch || exit 1; done

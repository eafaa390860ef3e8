R=$GRAFT_REPO_ROOT
run() { env "$@" timeout -k 10 120 python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', round(d['value']), 'Mrays/s')"; }
for rb in 4 8 16 24 32 48; do echo -n "refill_at_bounce=$rb: "; run RT_REFILL_AT_BOUNCE=$rb || exit 1; done

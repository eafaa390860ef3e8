R=$GRAFT_REPO_ROOT
run() { env "$@" timeout -k 10 120 python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', round(d['value']), 'Mrays/s')"; }
for ch in 256 512 1024 2048 4096; do echo -n "chunk=$ch: "; run RT_CHUNK=$ch || exit 1; done

#!/bin/bash
# Profile one bench.py run on the GPU box:   tools/profile.sh <tag> [bench args...]
# Writes gpurun_out/<tag>/{kernel_stats.csv, passes.txt, pmc_summary.txt, bench.json}; copy what should be kept to profiles/.
# Kernel trace and every counter pass are separate rocprofv3 runs (counter slots per pass: 8 SQ, 4 TCC).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-prof}; shift || true
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-also $*"

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/bench.json 2> $OUT/trace.log || { echo "kernel trace run failed"; tail -5 $OUT/trace.log; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 - "$OUT" <<'EOF'
import csv, glob, re, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(out + "/passes.txt", "w") as fh:
    fh.write("kernel duration_ms arch_vgpr accum_vgpr sgpr lds_bytes scratch_bytes grid block   (last frame of the run)\n")
    last = [r for r in rows if re.search(r"(intersect|shade|resolve|bounce|assemble)_kernel", r["Kernel_Name"])]
    n_res = [i for i, r in enumerate(last) if "resolve" in r["Kernel_Name"]]
    if len(n_res) >= 2:
        last = last[n_res[-2] + 1:]
    for r in last:
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", r["Kernel_Name"])
        fh.write(f"{m.group(1)}{m.group(2) or ''} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.3f} {r['VGPR_Count']} {r['Accum_VGPR_Count']} "
                 f"{r['SGPR_Count']} {r['LDS_Block_Size']} {r['Scratch_Size']} {r['Grid_Size_X']} {r['Workgroup_Size_X']}\n")
EOF
rm -rf $OUT/trace
P1="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU"
P2="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64"
P3="FETCH_SIZE TCC_HIT_sum"
P4="WRITE_SIZE TCC_MISS_sum"
P5="GRBM_GUI_ACTIVE SQ_WAVES SQ_LEVEL_WAVES"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i + 1))
    timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-also $* > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/pmc$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 > $OUT/pmc_summary.txt
rm -rf $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 $OUT/pmc*.log $OUT/trace.log
echo "--- $OUT/passes.txt"; cat $OUT/passes.txt
echo "--- $OUT/pmc_summary.txt"; grep -A40 -E "^(intersect|shade|bounce)" $OUT/pmc_summary.txt
